// Fused temporal focal window attention for gfx950 (fp32 MFMA, online softmax).
//
// The reference builds, per block, 4 rolled copies of K and V, window-partitions them, gathers the
// 120 "ring" positions, unfolds the pooled K/V, concatenates everything to [nWin*B, 4, T*210, 128]
// and materialises the [.., T*45, T*210] score tensor (242 MB at 432x240 T=10, 19.6 GB at 1080p
// T=20) -- model/modules/tfocal_transformer.py:226-396.  Here none of that exists: one workgroup
// owns (clip, window, head, block of query rows), walks the window's key list through a small
// per-window reference table (own tokens, circularly wrapped ring tokens incl. the 12 duplicates,
// valid pooled windows) and streams K/V rows of 512 B straight from the qkv GEMM output.
//
// MFMA mapping (v_mfma_f32_32x32x2_f32), one wave = 32 query rows:
//   S^T[key][query]  = K_tile . Q^T      A = K rows from LDS (b128 reads, 4 k per read),
//                                        B = the wave's Q rows, resident in 64 VGPRs (pre-scaled)
//   -> each lane holds 16 keys of ONE query: row max / row sum are in-register (+1 half swap)
//   O^T[d][query]   += V^T . P           B = P straight from the S^T registers (no shuffle),
//                                        A = V columns from LDS (conflict-free b32 reads)
//   -> per-query rescale and final 1/l are per-lane scalars.
// The zero-padded pooled slots score exactly -100 with V = 0 (reference :301-316,378-380): they are
// not multiplied, their exp(-100 - m) mass is added to the denominator analytically.
#include "common.h"

namespace {

constexpr int HD = 128;        // head dim
constexpr int NH = 4;          // heads
constexpr int CQ = 1536;       // qkv row length
constexpr int WS0 = 5, WS1 = 9, WTOK = 45;
constexpr int SLOTS = 210;     // key slots per frame: 45 own + 120 rolled + 45 pooled
constexpr int TK = 32;         // keys per tile
constexpr int LDK = HD + 4;    // padded LDS row
constexpr float LOG2E = 1.4426950408889634f;

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}

// Software pipeline per 32-key tile (LDS double-buffered, ONE barrier per tile):
//   issue K(t+1) loads -> S^T = K(t).Q^T (64 MFMA) -> write K(t+1) to the other LDS buffer -> issue V(t+1) loads
//   -> online softmax -> O^T += V(t)^T.P (64 MFMA) -> write V(t+1) -> barrier.
// The staging registers are shared by the K and V halves (32 VGPRs); rows past the key list are fetched with an
// out-of-range buffer offset and arrive as zeros.
// V is read with the head dim re-mapped d = 4*i + dt (i = MFMA row, dt = accumulator tile): one ds_read_b128 feeds the
// four PV MFMAs of a key and the epilogue stores 4 consecutive d per register.
// ONE_RSRC: qkv and kv_pool lie within one 4 GiB window (the engine allocates them back to back), so a single buffer
// resource based at the lower pointer serves both and a staging load is one instruction; otherwise two loads (one of
// them out of range) are summed.
// KS = 2: two key-groups of NW waves each take alternate key tiles (own LDS rings, own running max / sum / O) and are
// merged through LDS at the end -- two waves per SIMD when there are too few (clip, window, head, query block) items to
// give every CU more than one workgroup.
template <int NW, bool ONE_RSRC, int KS>
__global__ __launch_bounds__(64 * NW * KS) void focal_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ kvp,
                                                            const int* __restrict__ key_tab, int tab_ld,
                                                            const int* __restrict__ nkeys, float* __restrict__ out,
                                                            int B, int T, int fh, int fw, unsigned qkv_bytes,
                                                            unsigned kvp_bytes, const char* lo_base, unsigned lo_bytes,
                                                            unsigned q_rel, unsigned p_rel) {
    constexpr int NT = 64 * NW;
    constexpr int F4 = TK * 32;                 // float4s of one K (or V) tile
    constexpr int L_IT = (F4 + NT - 1) / NT;
    constexpr unsigned OOB = 0xFFFFFFFFu;
    __shared__ __attribute__((aligned(16))) float smem[KS * 4 * TK * LDK];
    __shared__ int stab[256];                   // the window's key table (<= 210 entries): looked up once per staged row,
                                                // from LDS -- a global lookup put a dependent L2 round trip in front of
                                                // every tile's K loads
    static_assert(KS == 1 || 4 * TK * LDK >= 66 * NT, "merge scratch must fit one group's rings");

    const int kg = (KS == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x / NT));
    const int tid = threadIdx.x - kg * NT;
    float* const sK0 = smem + kg * (4 * TK * LDK);
    float* const sV0 = sK0 + 2 * TK * LDK;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int nWw = fw / WS1, nWh = fh / WS0, nWin = nWh * nWw;
    const int win = blockIdx.y / NH, head = blockIdx.y - win * NH;
    const int wy = win / nWw, wx = win - wy * nWw;
    const int b = blockIdx.z;
    const int NQ = T * WTOK;
    const int ntok = fh * fw;

    const __amdgpu_buffer_rsrc_t r_qkv = ONE_RSRC ? make_rsrc(lo_base, lo_bytes) : make_rsrc(qkv, qkv_bytes);
    const __amdgpu_buffer_rsrc_t r_kvp = make_rsrc(kvp, kvp_bytes);

    // ---- this wave's 32 queries
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const bool wave_active = q0 < NQ;              // wave-uniform
    const int qi = q0 + i;
    const bool q_ok = qi < NQ;
    long long q_row = 0;
    {
        const int qq = q_ok ? qi : 0;
        const int t = qq / WTOK, pp = qq - t * WTOK;
        const int py = pp / WS1, px = pp - py * WS1;
        q_row = (long long)(b * T + t) * ntok + (wy * WS0 + py) * fw + (wx * WS1 + px);
    }
    const float qscale = 0.08838834764831845f * LOG2E;   // 128^-0.5 * log2(e)
    f32x4 q[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (q_ok) v = *reinterpret_cast<const f32x4*>(qkv + q_row * CQ + head * HD + 8 * m + 4 * h);
        q[m] = v * qscale;
    }

    f32x16 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    const int nv = nkeys[win];
    const int NK = T * nv;
    const int ntiles = (NK + TK - 1) / TK;
    const int* tab = key_tab + (long long)win * tab_ld;
    for (int e = threadIdx.x; e < nv && e < 256; e += KS * NT) stab[e] = tab[e];
    __syncthreads();

    // staging: thread handles float4 f = tid + it*NT  -> row f>>5, column chunk f&31
    f32x4 stg[L_IT];
    unsigned koff[L_IT];        // byte offset of this thread's K chunk (V = +2048), OOB for rows past the list
    bool kpool[L_IT];
    // (frame, slot) of this thread's rows in the group's next tile, advanced incrementally (no division per tile)
    int kt_t[L_IT], kt_s[L_IT];
#pragma unroll
    for (int it = 0; it < L_IT; ++it) {
        const int ks = kg * TK + ((tid + it * NT) >> 5);
        kt_t[it] = ks / nv;
        kt_s[it] = ks - kt_t[it] * nv;
    }
    auto tile_addresses = [&](int kt) {
#pragma unroll
        for (int it = 0; it < L_IT; ++it) {
            const int f = tid + it * NT;
            const int row = f >> 5, c = f & 31;
            const int ks = kt * TK + row;
            const bool ok = (F4 % NT == 0 || f < F4) && ks < NK;
            const int t = ok ? kt_t[it] : 0, s = ok ? kt_s[it] : 0;
            // next tile of this group is KS*TK rows further (nv >= 165 > KS*TK: at most one wrap)
            kt_s[it] += KS * TK;
            if (kt_s[it] >= nv) { kt_s[it] -= nv; kt_t[it] += 1; }
            const int ref = stab[s];
            const bool pooled = ref < 0;
            const unsigned rowi = pooled ? (unsigned)((b * T + t) * nWin + (-(ref + 1))) : (unsigned)((b * T + t) * ntok + ref);
            koff[it] = ok ? rowi * (unsigned)(CQ * 4) + (unsigned)((512 + head * HD + c * 4) * 4) +
                                (ONE_RSRC ? (pooled ? p_rel : q_rel) : 0u)
                          : OOB;
            kpool[it] = pooled;
        }
    };
    auto issue = [&](unsigned extra) {
#pragma unroll
        for (int it = 0; it < L_IT; ++it) {
            const unsigned o = koff[it] == OOB ? OOB : koff[it] + extra;
            if (ONE_RSRC) {
                stg[it] = buf_load4(r_qkv, o);
            } else {
                const f32x4 a = buf_load4(r_qkv, kpool[it] ? OOB : o);
                const f32x4 c = buf_load4(r_kvp, kpool[it] ? o : OOB);
                stg[it] = a + c;       // exactly one of the two is non-zero
            }
        }
    };
    auto commit = [&](float* dst) {
#pragma unroll
        for (int it = 0; it < L_IT; ++it) {
            const int f = tid + it * NT;
            if (F4 % NT == 0 || f < F4) *reinterpret_cast<f32x4*>(dst + (f >> 5) * LDK + (f & 31) * 4) = stg[it];
        }
    };

    // prologue: this group's first tile staged synchronously
    tile_addresses(kg);
    issue(0u);
    commit(sK0);
    issue(2048u);
    commit(sV0);
    __syncthreads();

    int cur = 0;
    const int nIter = (ntiles + KS - 1) / KS;
    for (int itr = 0; itr < nIter; ++itr) {
        const int kt = kg + itr * KS;           // may be == ntiles for the last group: an all-masked tile, harmless
        const float* cK = sK0 + cur * (TK * LDK);
        const float* cV = sV0 + cur * (TK * LDK);
        tile_addresses(kt + KS);                // rows past the end -> OOB -> zeros
        issue(0u);                              // K(t+1) in flight during the S phase

        f32x16 s;
        if (wave_active) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(cK + i * LDK + (2 * m + h) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], q[m][e], s, 0, 0, 0);
            }
        }
        commit(sK0 + (cur ^ 1) * (TK * LDK));
        issue(2048u);                           // V(t+1) in flight during softmax + PV

        if (wave_active) {
            float mx = -1e30f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (kt * TK + krow >= NK) s[r] = -1e30f;
                mx = fmaxf(mx, s[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(s[r] - m_new);
                psum += s[r];
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
            // rescale the accumulators only when some query's running maximum moved: after the first tiles it rarely
            // does, alpha is then exactly 1 and the 64 multiplies (+ their accumulator-register moves) are skipped
            if (__any(alpha != 1.0f)) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[dt][r] *= alpha;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = (r & 3) + 8 * (r >> 2) + 4 * h;
                const f32x4 a = *reinterpret_cast<const f32x4*>(cV + krow * LDK + 4 * i);     // d = 4*i + dt
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dt], s[r], acc[dt], 0, 0, 0);
            }
        }
        commit(sV0 + (cur ^ 1) * (TK * LDK));
        __syncthreads();
        cur ^= 1;
    }

    if (KS > 1) {
        // merge the key-groups: O = sum_g 2^(m_g - m) O_g, l likewise (each lane's l is still its half-row partial)
        float* scr = smem + kg * (4 * TK * LDK);       // the group's own rings are free now
        if (kg > 0) {
            scr[tid] = m_run;
            scr[NT + tid] = l_run;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) scr[(2 + dt * 16 + r) * NT + tid] = acc[dt][r];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int g = 1; g < KS; ++g) {
            const float* sg = smem + g * (4 * TK * LDK);
            const float m_o = sg[tid], l_o = sg[NT + tid];
            const float m_new = fmaxf(m_run, m_o);
            const float a0 = __builtin_amdgcn_exp2f(m_run - m_new), a1 = __builtin_amdgcn_exp2f(m_o - m_new);
            l_run = l_run * a0 + l_o * a1;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[dt][r] = acc[dt][r] * a0 + sg[(2 + dt * 16 + r) * NT + tid] * a1;
        }
    }

    if (wave_active) {
        float l = l_run + __shfl_xor(l_run, 32);
        const float nmask = (float)(T * (SLOTS - nv));
        l += nmask * __builtin_amdgcn_exp2f(-100.f * LOG2E - m_run);
        const float inv = 1.f / l;
        if (q_ok) {
            float* op = out + q_row * (NH * HD) + head * HD;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;      // MFMA row i -> d = 4*i + dt
                f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                *reinterpret_cast<f32x4*>(op + 4 * row) = v * inv;
            }
        }
    }
}

}  // namespace

extern "C" int e2fgvi_focal_attention(const float* qkv, const float* kv_pool, const int32_t* key_tab, int32_t tab_ld,
                                      const int32_t* nkeys, float* out, int32_t B, int32_t T, int32_t fh, int32_t fw,
                                      int32_t waves, void* stream) {
    E2_REQUIRE(qkv && kv_pool && key_tab && nkeys && out, E2FGVI_EINVAL, "focal_attention: null pointer");
    E2_REQUIRE(B > 0 && T > 0 && fh > 0 && fw > 0 && fh % WS0 == 0 && fw % WS1 == 0, E2FGVI_EINVAL,
               "focal_attention: token grid %dx%d must be a positive multiple of (5,9)", fh, fw);
    E2_REQUIRE(tab_ld >= SLOTS, E2FGVI_EINVAL, "focal_attention: tab_ld < 210");
    E2_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)kv_pool & 15) == 0 && ((uintptr_t)out & 15) == 0, E2FGVI_EINVAL,
               "focal_attention: buffers must be 16-byte aligned");
    const int qtiles = cdiv(T * WTOK, 32);
    // `waves`: 0 = auto; 2 or 4 = query waves per workgroup; +10 forces two key-groups (e.g. 14 = 4 waves x 2 groups)
    int ks = 0;
    if (waves >= 10) { ks = 2; waves -= 10; }
    if (waves <= 0) waves = qtiles <= 2 ? 2 : 4;
    E2_REQUIRE(waves == 2 || waves == 4, E2FGVI_EINVAL, "focal_attention: waves must be 0, 2, 4, 12 or 14");
    const int nWin = (fh / WS0) * (fw / WS1);
    const long long qb = (long long)B * T * fh * fw * CQ * 4, pb = (long long)B * T * nWin * CQ * 4;
    E2_REQUIRE(qb < 4294967295LL, E2FGVI_EUNSUP, "focal_attention: qkv spans >= 4 GiB; split the batch");
    dim3 grid(cdiv(qtiles, waves), nWin * NH, B), block(64 * waves);
    hipStream_t st = (hipStream_t)stream;
    const char* cq = (const char*)qkv;
    const char* cp = (const char*)kv_pool;
    const char* lo = cq < cp ? cq : cp;
    const long long hi_end = (cq + qb > cp + pb ? cq + qb : cp + pb) - lo;
    const bool one = hi_end < 4294967295LL;
    const unsigned lo_bytes = one ? (unsigned)hi_end : 0u, q_rel = one ? (unsigned)(cq - lo) : 0u, p_rel = one ? (unsigned)(cp - lo) : 0u;
    if (ks == 0) ks = ((long long)grid.x * grid.y * grid.z < 384) ? 2 : 1;     // < 1.5 workgroups per CU: split the keys
    block = dim3(64 * waves * ks);
#define E2_ATT(NW_, ONE_, KS_)                                                                                        \
    hipLaunchKernelGGL((focal_attn_kernel<NW_, ONE_, KS_>), grid, block, 0, st, qkv, kv_pool, key_tab, tab_ld, nkeys,  \
                       out, B, T, fh, fw, (unsigned)qb, (unsigned)pb, lo, lo_bytes, q_rel, p_rel)
#define E2_ATT_W(ONE_, KS_) do { if (waves == 2) E2_ATT(2, ONE_, KS_); else E2_ATT(4, ONE_, KS_); } while (0)
    if (one) { if (ks == 2) E2_ATT_W(true, 2); else E2_ATT_W(true, 1); }
    else     { if (ks == 2) E2_ATT_W(false, 2); else E2_ATT_W(false, 1); }
#undef E2_ATT_W
#undef E2_ATT
    E2_LAUNCH_CHECK("focal_attention");
    return 0;
}
