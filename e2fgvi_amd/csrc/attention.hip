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

__global__ __launch_bounds__(512) void focal_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ kvp,
                                                         const int* __restrict__ key_tab, int tab_ld,
                                                         const int* __restrict__ nkeys, float* __restrict__ out,
                                                         int B, int T, int fh, int fw) {
    __shared__ __attribute__((aligned(16))) float sK[TK * LDK];
    __shared__ __attribute__((aligned(16))) float sV[TK * LDK];

    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int nWw = fw / WS1, nWh = fh / WS0, nWin = nWh * nWw;
    const int win = blockIdx.y / NH, head = blockIdx.y - win * NH;
    const int wy = win / nWw, wx = win - wy * nWw;
    const int b = blockIdx.z;
    const int NQ = T * WTOK;
    const int ntok = fh * fw;

    // ---- this wave's 32 queries
    const int q0 = (blockIdx.x * nwave + wave) * 32;
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

    for (int kt = 0; kt < ntiles; ++kt) {
        // ---- stage K and V rows of this tile (32 rows x 32 float4 each)
        for (int f = tid; f < TK * 32; f += nthr) {
            const int row = f >> 5, c = f & 31;
            const int ks = kt * TK + row;
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (ks < NK) {
                const int t = ks / nv, s = ks - t * nv;
                const int ref = tab[s];
                const float* rp = (ref >= 0) ? qkv + ((long long)(b * T + t) * ntok + ref) * CQ
                                             : kvp + ((long long)(b * T + t) * nWin + (-(ref + 1))) * CQ;
                kv = *reinterpret_cast<const f32x4*>(rp + 512 + head * HD + c * 4);
                vv = *reinterpret_cast<const f32x4*>(rp + 1024 + head * HD + c * 4);
            }
            *reinterpret_cast<f32x4*>(sK + row * LDK + c * 4) = kv;
            *reinterpret_cast<f32x4*>(sV + row * LDK + c * 4) = vv;
        }
        __syncthreads();

        if (wave_active) {
            // ---- S^T = K . Q^T
            f32x16 s;
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(sK + i * LDK + (2 * m + h) * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], q[m][e], s, 0, 0, 0);
            }
            // ---- online softmax (per query = per lane pair l, l^32)
            float mx = -1e30f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = (r & 3) + 8 * (r >> 2) + 4 * h;
                if (kt * TK + krow >= NK) s[r] = -1e30f;
                mx = fmaxf(mx, s[r]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run, mx);
            const float alpha = exp2f(m_run - m_new);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = exp2f(s[r] - m_new);
                psum += s[r];
            }
            l_run = l_run * alpha + psum;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[dt][r] *= alpha;
            // ---- O^T += V^T . P
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int krow = (r & 3) + 8 * (r >> 2) + 4 * h;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const float a = sV[krow * LDK + dt * 32 + i];
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, s[r], acc[dt], 0, 0, 0);
                }
            }
        }
        __syncthreads();
    }

    if (wave_active) {
        float l = l_run + __shfl_xor(l_run, 32);
        const float nmask = (float)(T * (SLOTS - nv));
        l += nmask * exp2f(-100.f * LOG2E - m_run);
        const float inv = 1.f / l;
        if (q_ok) {
            float* op = out + q_row * (NH * HD) + head * HD;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    f32x4 v = {acc[dt][4 * rq + 0], acc[dt][4 * rq + 1], acc[dt][4 * rq + 2], acc[dt][4 * rq + 3]};
                    *reinterpret_cast<f32x4*>(op + dt * 32 + 8 * rq + 4 * h) = v * inv;
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
    if (waves <= 0) {
        // pick the wave count (2..6) that wastes the fewest wave tiles
        int best = 4, waste = 1 << 30;
        for (int w = 6; w >= 2; --w) {
            const int ws = cdiv(qtiles, w) * w - qtiles;
            if (ws < waste) { waste = ws; best = w; }
        }
        waves = best;
    }
    E2_REQUIRE(waves >= 1 && waves <= 8, E2FGVI_EINVAL, "focal_attention: waves must be in 1..8");
    const int nWin = (fh / WS0) * (fw / WS1);
    dim3 grid(cdiv(qtiles, waves), nWin * NH, B), block(64 * waves);
    hipLaunchKernelGGL(focal_attn_kernel, grid, block, 0, (hipStream_t)stream, qkv, kv_pool, key_tab, tab_ld, nkeys, out,
                       B, T, fh, fw);
    E2_LAUNCH_CHECK("focal_attention");
    return 0;
}
