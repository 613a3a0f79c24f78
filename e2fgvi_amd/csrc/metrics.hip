// PSNR / SSIM of evaluate.py (core/metrics.py:20-56) on the device, fp64 like the reference (img.astype(np.float64)):
//   psnr = 20 log10(255 / sqrt(mean((a-b)^2)))                                   (inf when identical)
//   ssim = skimage compare_ssim(a, b, data_range=255, multichannel=True, win_size=65): per channel, uniform 65x65 window,
//          sample covariance (NP/(NP-1)), K1 = 0.01, K2 = 0.03, S averaged over the image cropped by 32 pixels per side
//          (there every window lies inside the image, so the filter's border mode never matters), then over channels.
// Box sums come from fp64 summed-area tables: frames are uint8-valued (or k/2^j after the 0.5/0.5 blends), so every
// partial sum is exact in fp64 -- the only rounding is in the final SSIM arithmetic.  Reductions are two-stage with a
// fixed order (deterministic).
#include "common.h"

namespace {

constexpr int NTH = 256;
inline unsigned blocks_for(long long n) { return (unsigned)((n + NTH - 1) / NTH); }

// table layout: T[n][q][c][(H+1)][(W+1)], q = 0..4 for a, b, a*a, b*b, a*b; row 0 and column 0 are zero
__global__ void sat_rows_kernel(const float* __restrict__ a, const float* __restrict__ b, double* __restrict__ T, int N, int H, int W) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // (n, c, y)
    if (idx >= (long long)N * 3 * (H + 1)) return;
    const int y = (int)(idx % (H + 1));
    const int c = (int)((idx / (H + 1)) % 3);
    const int n = (int)(idx / ((long long)3 * (H + 1)));
    const long long plane = (long long)(H + 1) * (W + 1);
    double* t0 = T + (((long long)n * 5) * 3 + c) * plane + (long long)y * (W + 1);
    const long long qs = 3 * plane;
    for (int q = 0; q < 5; ++q) t0[q * qs] = 0.0;
    if (y == 0) {
        for (int x = 1; x <= W; ++x)
            for (int q = 0; q < 5; ++q) t0[q * qs + x] = 0.0;
        return;
    }
    const float* pa = a + (((long long)n * H + (y - 1)) * W) * 3 + c;
    const float* pb = b + (((long long)n * H + (y - 1)) * W) * 3 + c;
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0;
    for (int x = 0; x < W; ++x) {
        const double va = (double)pa[(long long)x * 3], vb = (double)pb[(long long)x * 3];
        s0 += va; s1 += vb; s2 += va * va; s3 += vb * vb; s4 += va * vb;
        t0[x + 1] = s0; t0[qs + x + 1] = s1; t0[2 * qs + x + 1] = s2; t0[3 * qs + x + 1] = s3; t0[4 * qs + x + 1] = s4;
    }
}

__global__ void sat_cols_kernel(double* __restrict__ T, int N, int H, int W) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // (n, q, c, x)
    if (idx >= (long long)N * 15 * (W + 1)) return;
    const int x = (int)(idx % (W + 1));
    const long long pl = idx / (W + 1);
    double* t = T + pl * (long long)(H + 1) * (W + 1) + x;
    double s = 0.0;
    for (int y = 1; y <= H; ++y) {
        s += t[(long long)y * (W + 1)];
        t[(long long)y * (W + 1)] = s;
    }
}

__device__ __forceinline__ double block_sum(double v, double* red) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0)
        for (int i = 0; i < NTH / 64; ++i) s += red[i];
    __syncthreads();
    return s;          // valid in thread 0
}

// per-block partial sums: part[n][0][block] = sum of S over the block's cropped pixels (all channels), part[n][1][block] = sum (a-b)^2
__global__ void ssim_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, const double* __restrict__ T,
                                    double* __restrict__ part, int H, int W, int win, int nblk) {
    __shared__ double red[NTH / 64];
    const int n = blockIdx.y;
    const int pad = (win - 1) / 2;
    const int Hc = H - 2 * pad, Wc = W - 2 * pad;
    const long long ncrop = (long long)3 * Hc * Wc, nall = (long long)3 * H * W;
    const long long plane = (long long)(H + 1) * (W + 1);
    const double NP = (double)win * win, cov_norm = NP / (NP - 1.0);
    const double C1 = (0.01 * 255.0) * (0.01 * 255.0), C2 = (0.03 * 255.0) * (0.03 * 255.0);
    double ssum = 0.0, dsum = 0.0;
    for (long long i = (long long)blockIdx.x * NTH + threadIdx.x; i < nall; i += (long long)nblk * NTH) {
        const double d = (double)a[(long long)n * nall + i] - (double)b[(long long)n * nall + i];
        dsum += d * d;
        if (i < ncrop) {
            const int c = (int)(i % 3);
            const int x = (int)((i / 3) % Wc);
            const int y = (int)(i / ((long long)3 * Wc));
            // window rows [y, y + win), columns [x, x + win) of the image (centre (y + pad, x + pad))
            double m[5];
            for (int q = 0; q < 5; ++q) {
                const double* t = T + (((long long)n * 5 + q) * 3 + c) * plane;
                const double s = t[(long long)(y + win) * (W + 1) + x + win] - t[(long long)y * (W + 1) + x + win] -
                                 t[(long long)(y + win) * (W + 1) + x] + t[(long long)y * (W + 1) + x];
                m[q] = s / NP;
            }
            const double ux = m[0], uy = m[1];
            const double vx = cov_norm * (m[2] - ux * ux), vy = cov_norm * (m[3] - uy * uy), vxy = cov_norm * (m[4] - ux * uy);
            const double A1 = 2.0 * ux * uy + C1, A2 = 2.0 * vxy + C2, B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
            ssum += (A1 * A2) / (B1 * B2);
        }
    }
    const double s = block_sum(ssum, red);
    const double d = block_sum(dsum, red);
    if (threadIdx.x == 0) {
        part[((long long)n * 2 + 0) * nblk + blockIdx.x] = s;
        part[((long long)n * 2 + 1) * nblk + blockIdx.x] = d;
    }
}

__global__ void metrics_final_kernel(const double* __restrict__ part, double* __restrict__ out, int N, int H, int W, int win, int nblk) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    double s = 0.0, d = 0.0;
    for (int i = 0; i < nblk; ++i) { s += part[((long long)n * 2 + 0) * nblk + i]; d += part[((long long)n * 2 + 1) * nblk + i]; }
    const int pad = (win - 1) / 2;
    const double mse = d / ((double)3 * H * W);
    out[2 * n + 0] = mse == 0.0 ? (double)INFINITY : 20.0 * log10(255.0 / sqrt(mse));
    out[2 * n + 1] = s / ((double)3 * (H - 2 * pad) * (W - 2 * pad));
}

constexpr int PART_BLOCKS = 256;

}  // namespace

extern "C" int64_t e2fgvi_psnr_ssim_workspace(int32_t N, int32_t H, int32_t W) {
    if (N <= 0 || H <= 0 || W <= 0) { e2fgvi_set_error("psnr_ssim_workspace: bad sizes"); return E2FGVI_EINVAL; }
    return ((int64_t)N * 15 * (H + 1) * (W + 1) + (int64_t)N * 2 * PART_BLOCKS) * 8;
}

extern "C" int e2fgvi_psnr_ssim(const float* img1, const float* img2, int32_t N, int32_t H, int32_t W, int32_t win_size,
                                void* workspace, double* out, void* stream) {
    E2_REQUIRE(img1 && img2 && workspace && out && N > 0 && H > 0 && W > 0, E2FGVI_EINVAL, "psnr_ssim: bad arguments");
    E2_REQUIRE(win_size >= 3 && (win_size & 1) && win_size <= H && win_size <= W, E2FGVI_EINVAL,
               "psnr_ssim: win_size must be odd, >= 3 and <= min(H, W)");
    E2_REQUIRE(((uintptr_t)workspace & 7) == 0, E2FGVI_EINVAL, "psnr_ssim: workspace must be 8-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    double* T = (double*)workspace;
    double* part = T + (long long)N * 15 * (H + 1) * (W + 1);
    hipLaunchKernelGGL(sat_rows_kernel, dim3(blocks_for((long long)N * 3 * (H + 1))), dim3(NTH), 0, st, img1, img2, T, N, H, W);
    hipLaunchKernelGGL(sat_cols_kernel, dim3(blocks_for((long long)N * 15 * (W + 1))), dim3(NTH), 0, st, T, N, H, W);
    hipLaunchKernelGGL(ssim_partial_kernel, dim3(PART_BLOCKS, N, 1), dim3(NTH), 0, st, img1, img2, T, part, H, W, win_size, PART_BLOCKS);
    hipLaunchKernelGGL(metrics_final_kernel, dim3(blocks_for(N)), dim3(NTH), 0, st, part, out, N, H, W, win_size, PART_BLOCKS);
    E2_LAUNCH_CHECK("psnr_ssim");
    return 0;
}
