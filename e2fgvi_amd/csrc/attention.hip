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
#include <stdlib.h>

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


// ------------------------------------------------------------------------------------------------------------------------
// Round 3: the same operator with the staging taken out of the waves' instruction streams (the bf16 kernel's round-3
// structure, attention_bf16.hip).  At 432x240 T=10 there are 960 (window, head, query block) wave-tasks for 1024 SIMDs:
// each wave is alone on its SIMD (or shares it with its key-group partner, KS = 2), so every instruction it spends on
// gather addresses, staging registers and LDS stores is matrix-pipe idle time (round 2: 69 % MFMA-busy).  Here
//   * the byte offset of every key row of the window is computed once per workgroup into an LDS table;
//   * K and V tiles go global -> LDS by LDS-DMA (1 KiB = two 512-byte key rows per wave instruction, issued through inline
//     asm so that hipcc places no vmcnt wait inside the tile loop); the conflict-free K image is produced on the source
//     side (lane (row, slot) fetches chunk slot ^ (row & 15)), V stays row-major (its b128 reads are contiguous);
//   * operand read addresses are one base register each (K: base ^ 32 m; V: base + immediate).
// Arithmetic, key order and summation order are those of focal_attn_kernel (the two agree to ~1e-6: hipcc contracts the
// softmax of each instantiation on its own).
typedef __attribute__((address_space(3))) void a2_lds_void;
typedef int a2_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void a2_dma16(a2_i32x4 rsrc, unsigned lds_dst, unsigned voff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ a2_i32x4 a2_rsrc_words(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    a2_i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xFFFFu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
constexpr int A2_TILE = TK * HD * 4;            // 16 KB: one K (or V) tile, 512-byte rows
constexpr int A2_RING = 4 * A2_TILE;            // K[2] + V[2] of one key group

template <int NW, int KS>
__global__ __launch_bounds__(64 * NW * KS) void focal_attn_v2_kernel(const float* __restrict__ qkv, const int* __restrict__ key_tab,
                                                                     int tab_ld, const int* __restrict__ nkeys,
                                                                     float* __restrict__ out, int B, int T, int fh, int fw,
                                                                     const char* lo_base, unsigned lo_bytes, unsigned q_rel,
                                                                     unsigned p_rel) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NT = 64 * NW;                    // threads of one key group
    constexpr int PIECES = 16 / NW;                // 1-KiB DMA pieces of a K (and of a V) tile per wave
    constexpr unsigned OOB = 0xFFFFF000u;          // stays out of range after the in-row offset (< 4 KiB) is added
    static_assert(NW == 2 || NW == 4, "query waves per key group");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[KS * A2_RING + (KS > 1 ? 2048 : 0)];
    __shared__ int stab[256];
    extern __shared__ __attribute__((aligned(16))) unsigned ktab[];       // byte offset of every key row, OOB past the end

    const int kg = (KS == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x / NT));
    const int tid = threadIdx.x - kg * NT;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int nWw = fw / WS1, nWh = fh / WS0, nWin = nWh * nWw;
    // 1-D grid, query chunk fastest (an XCD-aware order that keeps the chunks of one (window, head) on one L2 measured
    // neutral here and slower in the bf16 kernel, attention_bf16.hip)
    const int nqc = (T * WTOK + 32 * NW - 1) / (32 * NW);
    const int logical = (int)blockIdx.x;
    const int qchunk = logical % nqc;
    const int wh = (logical / nqc) % (nWin * NH);
    const int b = logical / (nqc * nWin * NH);
    const int win = wh / NH, head = wh - win * NH;
    const int wy = win / nWw, wx = win - wy * nWw;
    const int NQ = T * WTOK;
    const int ntok = fh * fw;
    const a2_i32x4 rsrc = a2_rsrc_words(lo_base, lo_bytes);

    const int nv = nkeys[win];
    const int NK = T * nv;
    const int ntiles = (NK + TK - 1) / TK;
    const int nIter = (ntiles + KS - 1) / KS;
    const int* tab = key_tab + (long long)win * tab_ld;
    for (int e = threadIdx.x; e < nv && e < 256; e += KS * NT) stab[e] = tab[e];
    __syncthreads();
    {   // the key-row table: entry k = (frame t = k / nv, slot s = k % nv), walked without divisions; KS * nIter tiles
        int t = 0, sl = threadIdx.x;
        while (sl >= nv) { sl -= nv; ++t; }
        const unsigned head_off = (unsigned)((512 + head * HD) * 4);
        for (int k = threadIdx.x; k < KS * nIter * TK; k += KS * NT) {
            unsigned e = OOB;
            if (k < NK) {
                const int ref = stab[sl];
                const bool pooled = ref < 0;
                const unsigned rowi = pooled ? (unsigned)((b * T + t) * nWin + (-(ref + 1))) : (unsigned)((b * T + t) * ntok + ref);
                e = rowi * (unsigned)(CQ * 4) + head_off + (pooled ? p_rel : q_rel);
            }
            ktab[k] = e;
            sl += KS * NT;
            while (sl >= nv) { sl -= nv; ++t; }
        }
    }
    __syncthreads();

    // ---- this wave's 32 queries
    const int q0 = (qchunk * NW + wave) * 32;
    const bool wave_active = q0 < NQ;              // wave-uniform
    auto query_row = [&](bool& ok) -> long long {
        const int qi = q0 + i;
        ok = qi < NQ;
        const int qq = ok ? qi : 0;
        const int t = qq / WTOK, pp = qq - t * WTOK;
        const int py = pp / WS1, px = pp - py * WS1;
        return (long long)(b * T + t) * ntok + (wy * WS0 + py) * fw + (wx * WS1 + px);
    };
    const float qscale = 0.08838834764831845f * LOG2E;   // 128^-0.5 * log2(e)
    f32x4 q[16];
    {
        bool ok;
        const long long row = query_row(ok);
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(qkv + row * CQ + head * HD + 8 * m + 4 * h);
            q[m] = v * qscale;
        }
    }
    f32x16 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    // ---- DMA bookkeeping: piece pc = wave + NW * jp covers key rows 2 pc, 2 pc + 1 of a tile; lane -> (row, 16-byte slot)
    const int d_row0 = 2 * wave + (lane >> 5);                                   // + 2 NW jp
    const unsigned d_slot = (unsigned)(lane & 31);
    unsigned d_kb[PIECES];
#pragma unroll
    for (int jp = 0; jp < PIECES; ++jp) d_kb[jp] = (d_slot ^ (unsigned)((d_row0 + 2 * NW * jp) & 15)) * 16u;
    const unsigned d_vb = 2048u + d_slot * 16u;                                  // V sits 512 floats behind K
    const unsigned ring_lds = (unsigned)(unsigned long long)(a2_lds_void*)smem + (unsigned)(kg * A2_RING);
    auto issue_tile = [&](int kt, int stage) {
        const unsigned sk = ring_lds + (unsigned)(stage * A2_TILE + wave * 1024);
        const unsigned sv = ring_lds + (unsigned)(2 * A2_TILE + stage * A2_TILE + wave * 1024);
        unsigned e[PIECES];
#pragma unroll
        for (int jp = 0; jp < PIECES; ++jp) e[jp] = ktab[kt * TK + d_row0 + 2 * NW * jp];
#pragma unroll
        for (int jp = 0; jp < PIECES; ++jp) {
            a2_dma16(rsrc, __builtin_amdgcn_readfirstlane(sk + jp * NW * 1024), e[jp] + d_kb[jp]);
            a2_dma16(rsrc, __builtin_amdgcn_readfirstlane(sv + jp * NW * 1024), e[jp] + d_vb);
        }
    };
    // operand reads: K chunk 2 m + h of key i sits in slot (2 m + h) ^ (i & 15) = (2 m) ^ (h ^ (i & 15)); V row-major
    const int k_base = i * 512 + ((h ^ (i & 15)) << 4);
    const int v_base = 4 * h * 512 + 16 * i;

    // the Q rows are compiler-counted loads: make hipcc wait for them here, not inside the tile loop (attention_bf16.hip)
#pragma unroll
    for (int m = 0; m < 16; ++m) asm volatile("" : "+v"(q[m]));

    issue_tile(kg, 0);                            // tile numbers past the list read OOB table entries -> zero rows
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const unsigned char* const ring = smem + kg * A2_RING;
    int cur = 0;
    for (int itr = 0; itr < nIter; ++itr) {
        const int kt = kg + itr * KS;           // may be == ntiles for the last group: an all-masked tile, harmless
        const unsigned char* cK = ring + cur * A2_TILE;
        const unsigned char* cV = ring + 2 * A2_TILE + cur * A2_TILE;
        if (itr + 1 < nIter) issue_tile(kt + KS, cur ^ 1);

        if (wave_active) {
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(cK + (k_base ^ (32 * m)));
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], q[m][e], s, 0, 0, 0);
            }
            if ((kt + 1) * TK > NK) {             // only the last tile(s) hold rows past the key list
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int krow = (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (kt * TK + krow >= NK) s[r] = -1e30f;
                }
            }
            float mx = -1e30f;
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
            {
                const unsigned mb = __builtin_bit_cast(unsigned, mx);
                const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
            }
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
            if (__any(alpha != 1.0f)) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[dt][r] *= alpha;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(cV + v_base + ((r & 3) + 8 * (r >> 2)) * 512);     // d = 4*i + dt
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[dt], s[r], acc[dt], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of the next tile have landed
        __syncthreads();
        cur ^= 1;
    }

    if (KS > 1) {
        // merge the key-groups: O = sum_g 2^(m_g - m) O_g, l likewise (each lane's l is still its half-row partial)
        float* scr = reinterpret_cast<float*>(smem + kg * A2_RING);          // the group's own rings are free now (+ 2 KiB tail)
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
            const float* sg = reinterpret_cast<const float*>(smem + g * A2_RING);
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
        bool ok;
        const long long row = query_row(ok);
        if (ok) {
            float* op = out + row * (NH * HD) + head * HD;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int orow = (r & 3) + 8 * (r >> 2) + 4 * h;      // MFMA row i -> d = 4*i + dt
                f32x4 v = {acc[0][r], acc[1][r], acc[2][r], acc[3][r]};
                *reinterpret_cast<f32x4*>(op + 4 * orow) = v * inv;
            }
        }
    }
#endif
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
    // 24 / 22 = the LDS-DMA kernel (round 3) with one key group, 34 / 32 with two; 0 picks it whenever it applies
    int ks = 0, v2 = -1;
    if (waves >= 20) { v2 = 1; ks = waves >= 30 ? 2 : 1; waves -= ks == 2 ? 30 : 20; }
    else if (waves > 0) v2 = 0;
    if (waves >= 10) { ks = 2; waves -= 10; }
    if (waves <= 0) waves = qtiles <= 2 ? 2 : 4;
    E2_REQUIRE(waves == 2 || waves == 4, E2FGVI_EINVAL, "focal_attention: waves must be 0, 2, 4, 12, 14, 22, 24, 32 or 34");
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
    {
        const size_t dyn = (size_t)cdiv(T * SLOTS, TK * ks) * ks * TK * 4;
        const bool fits = one && hi_end < 0xFFFFF000LL && dyn + (size_t)ks * A2_RING + 2048 + 1024 + 256 <= 160 * 1024;
        if (v2 < 0) v2 = fits ? 1 : 0;
        if (v2 == 1 && !fits && ks == 2) {                  // the two-group rings leave no room for the table: one group
            const size_t dyn1 = (size_t)cdiv(T * SLOTS, TK) * TK * 4;
            if (one && hi_end < 0xFFFFF000LL && dyn1 + A2_RING + 1024 + 256 <= 160 * 1024) { ks = 1; block = dim3(64 * waves); }
            else v2 = 0;
        } else if (v2 == 1 && !fits) v2 = 0;
        if (v2 == 1) {
            const size_t dyn2 = (size_t)cdiv(T * SLOTS, TK * ks) * ks * TK * 4;
#define E2_ATT2(NW_, KS_)                                                                                                          \
            do {                                                                                                                   \
                static size_t reserved = 0;                                                                                        \
                if (dyn2 > reserved) {                                                                                             \
                    hipError_t ea = hipFuncSetAttribute((const void*)focal_attn_v2_kernel<NW_, KS_>,                               \
                                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn2);                     \
                    E2_REQUIRE(ea == hipSuccess, (int)ea, "focal_attention: cannot reserve %zu bytes of dynamic LDS", dyn2);       \
                    reserved = dyn2;                                                                                               \
                }                                                                                                                  \
                hipLaunchKernelGGL((focal_attn_v2_kernel<NW_, KS_>), dim3(grid.x * grid.y * grid.z), block, dyn2, st, qkv, key_tab, tab_ld, nkeys, out, B, T, \
                                   fh, fw, lo, lo_bytes, q_rel, p_rel);                                                            \
            } while (0)
            if (waves == 2) { if (ks == 2) E2_ATT2(2, 2); else E2_ATT2(2, 1); }
            else            { if (ks == 2) E2_ATT2(4, 2); else E2_ATT2(4, 1); }
#undef E2_ATT2
            E2_LAUNCH_CHECK("focal_attention (v2)");
            return 0;
        }
    }
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
