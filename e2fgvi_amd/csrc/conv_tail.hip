// The decoder's last layer (model/e2fgvi.py:99-103, model/e2fgvi_hq.py:99-103: nn.Conv2d(64, 3, kernel_size=3, stride=1,
// padding=1) followed by torch.tanh in InpaintGenerator.forward) for gfx950.
//
// A 3x3 layer with 3 output channels wastes 29 of the 32 columns of every MFMA when it is run as an implicit GEMM with
// K = 9 * 64, and its im2col reads every input pixel nine times.  Here the taps move to the N side instead:
//
//     Z[q][(tap, co)] = sum_c x[q][c] * w[co][c][tap]            one [pixels x 64] x [64 x 27] GEMM, every pixel read ONCE
//     y[p][co]        = bias[co] + sum_tap Z[p + delta(tap)][(tap, co)]
//
// 27 of 32 MFMA columns carry work, the K dimension is 64 instead of 576 (9x fewer matrix instructions), and the shifted sum
// is nine LDS reads per output value.  The kernel is HBM-bound by construction: the input tile (+ a one-pixel halo) is
// read once, straight from global memory into the MFMA A operand (no LDS staging: lane (pixel, half) owns 32 consecutive
// channels, and the K order inside a 64-channel row is free as long as the packed weights agree), the result is written
// once as fp32 NCHW frames.
//
// Workgroup: 256 threads, 16 x 32 output pixels; Z covers the 18 x 34 halo block = 612 pixels = 20 MFMA row blocks (5 per
// wave) x 27 floats in LDS (69 KB, two workgroups per CU).
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int TH = 16, TW = 32, ZH = TH + 2, ZW = TW + 2, ZPIX = ZH * ZW;   // 612
constexpr int NBLK = 5;                    // MFMA row blocks (32 pixels) per wave: 4 waves x 5 x 32 = 640 >= 612
constexpr int ZS = 27;                     // floats per Z pixel (odd: the per-pixel reads of the sum are conflict-free)

struct TailParams {
    const void* src;                       // NHWC [N, H, W, src_ld] bf16 / fp32, 64 channels used
    const void* wp;                        // packed weights (pack_tail_weight_kernel)
    const float* bias;                     // [3] or null
    float* dst;                            // NCHW fp32 [N, 3, H, W]
    int N, H, W, src_ld, tilesX, tilesY, act;
    float slope;
};

// packed weights: the B operand of every MFMA of a row block, lane-linear.
//   bf16: [kk 0..3][lane][8]:  column n = lane & 31, channels (lane >> 5) * 32 + kk * 8 + e
//   fp32: [j 0..31][lane]:     column n = lane & 31, channel  (lane >> 5) * 32 + j
// column n = tap * 3 + co (n < 27), zero above.
template <typename T, bool F32>
__global__ void pack_tail_weight_kernel(const float* __restrict__ w, T* __restrict__ wp) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= 64 * 32) return;
    int lane, ch;
    if (F32) { lane = idx & 63; ch = (lane >> 5) * 32 + (idx >> 6); }
    else { lane = (idx >> 3) & 63; ch = (lane >> 5) * 32 + (idx >> 9) * 8 + (idx & 7); }
    const int n = lane & 31, tap = n / 3, co = n - tap * 3;
    wp[idx] = (T)(n < 27 ? w[(co * 64 + ch) * 9 + tap] : 0.f);
}

template <bool F32>
__global__ __launch_bounds__(256) void conv_tail_kernel(const TailParams p) {
    __shared__ float Z[4 * NBLK * 32 * ZS];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 31, h = lane >> 5;
    int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY, img = bid / p.tilesY;
    const int y0 = ty * TH - 1, x0 = tx * TW - 1;          // image position of Z pixel (0, 0)
    constexpr int ESZ = F32 ? 4 : 2;
    constexpr int NLD = F32 ? 8 : 4;                       // 16-byte loads per lane and row block (32 channels)
    constexpr int DEPTH = F32 ? 2 : NBLK;                  // row blocks of loads in flight

    // B operand: the whole 64 x 32 weight matrix lives in registers
    f32x4 bw[NLD];
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        if constexpr (F32) {
            const float* w = reinterpret_cast<const float*>(p.wp);
#pragma unroll
            for (int e = 0; e < 4; ++e) bw[k][e] = w[(k * 4 + e) * 64 + lane];
        } else {
            bw[k] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const char*>(p.wp) + (k * 64 + lane) * 16);
        }
    }

    f32x4 fr[DEPTH][NLD];
    auto load = [&](f32x4 (&f)[NLD], int blk) {
        const int q = (wave * NBLK + blk) * 32 + i;
        const int zy = q / ZW, zx = q - zy * ZW;
        const int y = y0 + zy, x = x0 + zx;
        const bool ok = q < ZPIX && (unsigned)y < (unsigned)p.H && (unsigned)x < (unsigned)p.W;
        const char* s = reinterpret_cast<const char*>(p.src) +
                        ((((long long)img * p.H + (ok ? y : 0)) * p.W + (ok ? x : 0)) * p.src_ld + h * 32) * ESZ;
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(s + k * 16);
            f[k] = v;
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(fr[d], d);

#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        f32x4 (&f)[NLD] = fr[blk % DEPTH];
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            if constexpr (F32) {
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(f[k][e], bw[k][e], acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f[k]), __builtin_bit_cast(bf16x8, bw[k]),
                                                              acc, 0, 0, 0);
            }
        }
        if (blk + DEPTH < NBLK) load(fr[blk % DEPTH], blk + DEPTH);
        // D layout: register r of lane (i, h) = row (r & 3) + 8 (r >> 2) + 4 h, column i
        if (i < ZS) {
            float* z = Z + ((wave * NBLK + blk) * 32 + 4 * h) * ZS + i;
#pragma unroll
            for (int r = 0; r < 16; ++r) z[((r & 3) + 8 * (r >> 2)) * ZS] = acc[r];
        }
    }
    __syncthreads();

    // shifted sum: thread (oy, ox) of the 16 x 32 tile, two rows per thread
    const int ox = threadIdx.x & 31, x = tx * TW + ox;
    float b[3] = {0.f, 0.f, 0.f};
    if (p.bias) { b[0] = p.bias[0]; b[1] = p.bias[1]; b[2] = p.bias[2]; }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int oy = (threadIdx.x >> 5) + 8 * half, y = ty * TH + oy;
        float s[3] = {b[0], b[1], b[2]};
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float* z = Z + ((oy + ky) * ZW + ox + kx) * ZS + (ky * 3 + kx) * 3;
                s[0] += z[0]; s[1] += z[1]; s[2] += z[2];
            }
        if (y < p.H && x < p.W) {
            float* o = p.dst + (((long long)img * 3) * p.H + y) * p.W + x;
#pragma unroll
            for (int co = 0; co < 3; ++co) o[(long long)co * p.H * p.W] = apply_act(s[co], p.act, p.slope);
        }
    }
}

}  // namespace

/* see include/e2fgvi_hip.h */
extern "C" int64_t e2fgvi_packed_tail_weight_size(int32_t Cout, int32_t Cin) {
    if (Cout != 3 || Cin != 64) {
        e2fgvi_set_error("packed_tail_weight_size: the tail kernel is for 64 -> 3 channels (got %d -> %d)", Cin, Cout);
        return E2FGVI_EUNSUP;
    }
    return 64 * 32;
}

extern "C" int e2fgvi_pack_tail_weight(const float* w, void* wpacked, int32_t Cout, int32_t Cin, int32_t dtype, void* stream) {
    E2_REQUIRE(w && wpacked, E2FGVI_EINVAL, "pack_tail_weight: null pointer");
    E2_REQUIRE(Cout == 3 && Cin == 64, E2FGVI_EUNSUP, "pack_tail_weight: the tail kernel is for 64 -> 3 channels (got %d -> %d)", Cin, Cout);
    E2_REQUIRE(dtype == E2FGVI_F32 || dtype == E2FGVI_BF16, E2FGVI_EINVAL, "pack_tail_weight: dtype must be E2FGVI_F32 or E2FGVI_BF16");
    if (dtype == E2FGVI_F32)
        hipLaunchKernelGGL((pack_tail_weight_kernel<float, true>), dim3(8), dim3(256), 0, (hipStream_t)stream, w, (float*)wpacked);
    else
        hipLaunchKernelGGL((pack_tail_weight_kernel<__bf16, false>), dim3(8), dim3(256), 0, (hipStream_t)stream, w, (__bf16*)wpacked);
    E2_LAUNCH_CHECK("pack_tail_weight");
    return 0;
}

extern "C" int e2fgvi_conv3x3_tail(const void* src, int32_t src_dtype, int32_t src_ld, const void* wpacked, const float* bias,
                                   float* dst, int32_t N, int32_t H, int32_t W, int32_t act, float slope, void* stream) {
    E2_REQUIRE(src && wpacked && dst, E2FGVI_EINVAL, "conv3x3_tail: null pointer");
    E2_REQUIRE(src_dtype == E2FGVI_F32 || src_dtype == E2FGVI_BF16, E2FGVI_EINVAL, "conv3x3_tail: src_dtype must be E2FGVI_F32 or E2FGVI_BF16");
    E2_REQUIRE(N > 0 && H > 0 && W > 0 && src_ld >= 64, E2FGVI_EINVAL, "conv3x3_tail: bad shape");
    E2_REQUIRE(src_ld % (src_dtype == E2FGVI_F32 ? 4 : 8) == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)wpacked & 15) == 0, E2FGVI_EINVAL,
               "conv3x3_tail: source rows must be 16-byte aligned");
    E2_REQUIRE(act >= E2FGVI_ACT_NONE && act <= E2FGVI_ACT_TANH, E2FGVI_EINVAL, "conv3x3_tail: bad activation");
    TailParams p;
    p.src = src; p.wp = wpacked; p.bias = bias; p.dst = dst;
    p.N = N; p.H = H; p.W = W; p.src_ld = src_ld; p.act = act; p.slope = slope;
    p.tilesX = cdiv(W, TW); p.tilesY = cdiv(H, TH);
    const long long nblk = (long long)N * p.tilesX * p.tilesY;
    E2_REQUIRE(nblk < (1ll << 31), E2FGVI_EINVAL, "conv3x3_tail: too many tiles");
    if (src_dtype == E2FGVI_F32)
        hipLaunchKernelGGL((conv_tail_kernel<true>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL((conv_tail_kernel<false>), dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, p);
    E2_LAUNCH_CHECK("conv3x3_tail");
    return 0;
}
