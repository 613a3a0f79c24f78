// Fused temporal focal window attention of the bf16 data path (gfx950): bf16 qkv rows in HBM, both products on
// v_mfma_f32_32x32x16_bf16, fp32 online softmax, bf16 output.  Same operator, key tables and reference call sites as
// attention.hip (tfocal_transformer.py:226-396, tfocal_transformer_hq.py:231-425).
//
//   S^T[key][query] = K_tile . Q^T       A = K rows from LDS (one ds_read_b128 = 8 consecutive d), B = the wave's Q rows,
//                                        resident in 32 VGPRs; 8 MFMAs per 32-key tile
//   -> each lane holds 16 keys of ONE query: row max / row sum stay in registers (+1 half swap)
//   O^T[d][query]  += V^T . P            B = P straight from the S^T registers: the product sums over keys, so the MFMA's
//                                        k index is free to enumerate the keys in the order the lane already holds them
//                                        (k = 16 kk + 8 h + e  <->  key (r & 3) + 8 (r >> 2) + 4 h, r = 8 kk + e) -- no
//                                        cross-lane traffic for P;  A = V^T rows from LDS, where the V tile is stored
//                                        TRANSPOSED and in that same key order by the staging pass (2-byte scatter writes).
// LDS per workgroup (double buffered): K [32 keys][16 chunks of 8 d], chunk c of key k in slot c ^ (k & 15) (conflict-free
// b128 reads down a column of keys);  V^T [128 d][32 keys (+8 pad)], 16-byte unit u of row d in slot u ^ ((d >> 4) & 3)
// (80-byte rows + the XOR spread the transposing writes over the banks).
// Software pipeline per 32-key tile, ONE barrier per tile (as in attention.hip): issue K(t+1) loads -> S^T MFMAs ->
// write K(t+1) -> issue V(t+1) loads -> softmax -> PV MFMAs -> write V(t+1)^T -> barrier.
// The zero-padded pooled slots score exactly -100 with V = 0: their exp mass is added to the denominator analytically.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int HD = 128, NH = 4, CQ = 1536;
constexpr int WS0 = 5, WS1 = 9, WTOK = 45, SLOTS = 210;
constexpr int TK = 32;
constexpr int K_BYTES = TK * HD * 2;              // 8 KB
constexpr int VROW = 80;                          // bytes per V^T row: 32 keys * 2 + 16 pad
constexpr int V_BYTES = HD * VROW;                // 10 KB
constexpr float LOG2E = 1.4426950408889634f;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load4u(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
}

template <int NW>
__global__ __launch_bounds__(64 * NW) void focal_attn_bf16_kernel(const __bf16* __restrict__ qkv, const int* __restrict__ key_tab,
                                                                  int tab_ld, const int* __restrict__ nkeys,
                                                                  __bf16* __restrict__ out, int B, int T, int fh, int fw,
                                                                  const char* lo_base, unsigned lo_bytes, unsigned q_rel,
                                                                  unsigned p_rel) {
    constexpr int NT = 64 * NW;
    constexpr int ITEMS = TK * 16;                 // 16-byte items of one K (or V) tile
    constexpr int L_IT = ITEMS / NT;
    constexpr unsigned OOB = 0xFFFFFFFFu;
    static_assert(ITEMS % NT == 0, "tile items");
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * K_BYTES + 2 * V_BYTES];
    __shared__ int stab[256];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int nWw = fw / WS1, nWh = fh / WS0, nWin = nWh * nWw;
    const int win = blockIdx.y / NH, head = blockIdx.y - win * NH;
    const int wy = win / nWw, wx = win - wy * nWw;
    const int b = blockIdx.z;
    const int NQ = T * WTOK;
    const int ntok = fh * fw;
    const __amdgpu_buffer_rsrc_t rsrc = make_rsrc(lo_base, lo_bytes);

    // ---- this wave's 32 queries: 128 d as 8 operand octets per k-step
    const int q0 = (blockIdx.x * NW + wave) * 32;
    const bool wave_active = q0 < NQ;
    const int qi = q0 + i;
    const bool q_ok = qi < NQ;
    long long q_row = 0;
    {
        const int qq = q_ok ? qi : 0;
        const int t = qq / WTOK, pp = qq - t * WTOK;
        const int py = pp / WS1, px = pp - py * WS1;
        q_row = (long long)(b * T + t) * ntok + (wy * WS0 + py) * fw + (wx * WS1 + px);
    }
    bf16x8 q[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (q_ok) v = *reinterpret_cast<const u32x4*>(qkv + q_row * CQ + head * HD + kk * 16 + h * 8);
        q[kk] = __builtin_bit_cast(bf16x8, v);
    }
    const float qscale = 0.08838834764831845f * LOG2E;       // 128^-0.5 * log2(e), applied to the fp32 scores

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
    for (int e = tid; e < nv && e < 256; e += NT) stab[e] = tab[e];
    __syncthreads();

    // staging: item f = tid + it * NT -> key row f >> 4, 16-byte chunk (8 d) f & 15
    u32x4 stg[L_IT];
    unsigned koff[L_IT];
    int kt_t[L_IT], kt_s[L_IT];
#pragma unroll
    for (int it = 0; it < L_IT; ++it) {
        const int ks = (tid + it * NT) >> 4;
        kt_t[it] = ks / nv;
        kt_s[it] = ks - kt_t[it] * nv;
    }
    auto tile_addresses = [&](int kt) {
#pragma unroll
        for (int it = 0; it < L_IT; ++it) {
            const int f = tid + it * NT;
            const int row = f >> 4, c = f & 15;
            const bool ok = kt * TK + row < NK;
            const int t = ok ? kt_t[it] : 0, s = ok ? kt_s[it] : 0;
            kt_s[it] += TK;                                    // nv >= 165 > TK: at most one wrap per tile
            if (kt_s[it] >= nv) { kt_s[it] -= nv; kt_t[it] += 1; }
            const int ref = stab[s];
            const bool pooled = ref < 0;
            const unsigned rowi = pooled ? (unsigned)((b * T + t) * nWin + (-(ref + 1))) : (unsigned)((b * T + t) * ntok + ref);
            koff[it] = ok ? rowi * (unsigned)(CQ * 2) + (unsigned)((512 + head * HD + c * 8) * 2) + (pooled ? p_rel : q_rel) : OOB;
        }
    };
    auto issue = [&](unsigned extra) {
#pragma unroll
        for (int it = 0; it < L_IT; ++it) stg[it] = buf_load4u(rsrc, koff[it] == OOB ? OOB : koff[it] + extra);
    };
    auto commit_k = [&](unsigned char* dst) {
#pragma unroll
        for (int it = 0; it < L_IT; ++it) {
            const int f = tid + it * NT;
            const int row = f >> 4, c = f & 15;
            *reinterpret_cast<u32x4*>(dst + row * 256 + ((c ^ (row & 15)) << 4)) = stg[it];
        }
    };
    auto commit_vt = [&](unsigned char* dst) {
#pragma unroll
        for (int it = 0; it < L_IT; ++it) {
            const int f = tid + it * NT;
            const int key = f >> 4, c = f & 15;
            // position of this key in the PV product's k order (see the header): key = (r & 3) + 8 (r >> 2) + 4 hh
            const int hh = (key >> 2) & 1, r = (key >> 3) * 4 + (key & 3);
            const int kp = (r >> 3) * 16 + hh * 8 + (r & 7);
            const int unit = kp >> 3, within = (kp & 7) * 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = c * 8 + j;
                const unsigned w = stg[it][j >> 1];
                const unsigned short v = (j & 1) ? (unsigned short)(w >> 16) : (unsigned short)(w & 0xFFFFu);
                *reinterpret_cast<unsigned short*>(dst + d * VROW + ((unit ^ ((d >> 4) & 3)) << 4) + within) = v;
            }
        }
    };

    unsigned char* const sK0 = smem;
    unsigned char* const sV0 = smem + 2 * K_BYTES;
    tile_addresses(0);
    issue(0u);
    commit_k(sK0);
    issue(1024u);                                   // V is 512 bf16 = 1024 bytes behind K in a qkv row
    commit_vt(sV0);
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < ntiles; ++kt) {
        const unsigned char* cK = sK0 + cur * K_BYTES;
        const unsigned char* cV = sV0 + cur * V_BYTES;
        tile_addresses(kt + 1);                     // rows past the end -> OOB -> zeros
        issue(0u);

        f32x16 s;
        if (wave_active) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(cK + i * 256 + (((2 * kk + h) ^ (i & 15)) << 4));
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q[kk], s, 0, 0, 0);
            }
        }
        commit_k(sK0 + (cur ^ 1) * K_BYTES);
        issue(1024u);

        if (wave_active) {
            float mx = -1e30f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = (r & 3) + 8 * (r >> 2) + 4 * h;
                s[r] = (kt * TK + krow >= NK) ? -1e30f : s[r] * qscale;
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
            if (__any(alpha != 1.0f)) {
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[dt][r] *= alpha;
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 p;
#pragma unroll
                for (int e = 0; e < 8; ++e) p[e] = (__bf16)s[kk * 8 + e];
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int d = dt * 32 + i;
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(cV + d * VROW + (((2 * kk + h) ^ ((d >> 4) & 3)) << 4));
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, p, acc[dt], 0, 0, 0);
                }
            }
        }
        commit_vt(sV0 + (cur ^ 1) * V_BYTES);
        __syncthreads();
        cur ^= 1;
    }

    if (wave_active) {
        float l = l_run + __shfl_xor(l_run, 32);
        const float nmask = (float)(T * (SLOTS - nv));
        l += nmask * __builtin_amdgcn_exp2f(-100.f * LOG2E - m_run);
        const float inv = 1.f / l;
        if (q_ok) {
            __bf16* op = out + q_row * (NH * HD) + head * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 v = {(__bf16)(acc[dt][rq * 4 + 0] * inv), (__bf16)(acc[dt][rq * 4 + 1] * inv),
                                (__bf16)(acc[dt][rq * 4 + 2] * inv), (__bf16)(acc[dt][rq * 4 + 3] * inv)};
                    *reinterpret_cast<bf16x4*>(op + dt * 32 + 8 * rq + 4 * h) = v;      // MFMA rows (r&3) + 8 (r>>2) + 4 h
                }
        }
    }
}

}  // namespace

extern "C" int e2fgvi_focal_attention_bf16(const void* qkv, const void* kv_pool, const int32_t* key_tab, int32_t tab_ld,
                                           const int32_t* nkeys, void* out, int32_t B, int32_t T, int32_t fh, int32_t fw,
                                           void* stream) {
    E2_REQUIRE(qkv && kv_pool && key_tab && nkeys && out, E2FGVI_EINVAL, "focal_attention_bf16: null pointer");
    E2_REQUIRE(B > 0 && T > 0 && fh > 0 && fw > 0 && fh % WS0 == 0 && fw % WS1 == 0, E2FGVI_EINVAL,
               "focal_attention_bf16: token grid %dx%d must be a positive multiple of (5,9)", fh, fw);
    E2_REQUIRE(tab_ld >= SLOTS, E2FGVI_EINVAL, "focal_attention_bf16: tab_ld < 210");
    E2_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)kv_pool & 15) == 0 && ((uintptr_t)out & 15) == 0, E2FGVI_EINVAL,
               "focal_attention_bf16: buffers must be 16-byte aligned");
    const int qtiles = cdiv(T * WTOK, 32);
    const int nWin = (fh / WS0) * (fw / WS1);
    const long long qb = (long long)B * T * fh * fw * CQ * 2, pb = (long long)B * T * nWin * CQ * 2;
    const char* cq = (const char*)qkv;
    const char* cp = (const char*)kv_pool;
    const char* lo = cq < cp ? cq : cp;
    const long long hi_end = (cq + qb > cp + pb ? cq + qb : cp + pb) - lo;
    E2_REQUIRE(hi_end < 4294967295LL, E2FGVI_EUNSUP,
               "focal_attention_bf16: qkv and kv_pool must lie within one 4 GiB window (allocate them back to back / split the batch)");
    static int nw_env = -1;
    if (nw_env < 0) { const char* e = getenv("E2FGVI_ATT_NW"); nw_env = e ? atoi(e) : 0; }
    // eight query waves per workgroup when a window has enough query tiles (720p T=10: 15): every staged K / V tile then
    // serves 256 queries instead of 128, half the staging work per MFMA (-0.2 ms per 720p forward); E2FGVI_ATT_NW overrides
    const int nw = nw_env ? nw_env : (qtiles >= 12 ? 8 : 4);
    dim3 grid(cdiv(qtiles, nw), nWin * NH, B), block(64 * nw);
    if (nw == 8)
        hipLaunchKernelGGL(focal_attn_bf16_kernel<8>, grid, block, 0, (hipStream_t)stream, (const __bf16*)qkv, key_tab, tab_ld, nkeys,
                           (__bf16*)out, B, T, fh, fw, lo, (unsigned)hi_end, (unsigned)(cq - lo), (unsigned)(cp - lo));
    else if (nw == 2)
        hipLaunchKernelGGL(focal_attn_bf16_kernel<2>, grid, block, 0, (hipStream_t)stream, (const __bf16*)qkv, key_tab, tab_ld, nkeys,
                           (__bf16*)out, B, T, fh, fw, lo, (unsigned)hi_end, (unsigned)(cq - lo), (unsigned)(cp - lo));
    else
    hipLaunchKernelGGL(focal_attn_bf16_kernel<4>, grid, block, 0, (hipStream_t)stream, (const __bf16*)qkv, key_tab, tab_ld, nkeys,
                       (__bf16*)out, B, T, fh, fw, lo, (unsigned)hi_end, (unsigned)(cq - lo), (unsigned)(cp - lo));
    E2_LAUNCH_CHECK("focal_attention_bf16");
    return 0;
}
