// Byte-side kernels of the sliding-window video driver: everything test.py does to the
// uint8 frames and masks around the model call, on the device.  HBM-bound integer / byte work: one thread per output
// element, coalesced along the innermost (channel / x) index; results are bit-exact with the numpy / PIL reference.
#include "common.h"

namespace {

constexpr int NTH = 256;
inline unsigned blocks_for(long long n) { return (unsigned)((n + NTH - 1) / NTH); }

// Mask preparation (test.py:56-69): NEAREST resize to the frame size, binarise (> 0), 4x dilation with the 3x3 cross.
// The source row / column of every output row / column comes in as a table (Pillow builds the same tables in
// ImagingScaleAffine; the host mirrors its double arithmetic, e2fgvi_amd/video.py::nearest_table).
__global__ void mask_prepare_kernel(const unsigned char* __restrict__ src, int Hin, int Win, const int* __restrict__ ytab,
                                    const int* __restrict__ xtab, unsigned char* __restrict__ dst, int L, int H, int W, int iters) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)L * H * W) return;
    const int x = (int)(idx % W);
    const int y = (int)((idx / W) % H);
    const int l = (int)(idx / ((long long)W * H));
    const unsigned char* s = src + (long long)l * Hin * Win;
    // `iters` dilations with the 3x3 cross = OR over the L1 ball of that radius, clipped to the image
    // (cv2.dilate ignores out-of-image pixels; test.py:64-68)
    int hit = 0;
    for (int dy = -iters; dy <= iters && !hit; ++dy) {
        const int yy = y + dy;
        if (yy < 0 || yy >= H) continue;
        const long long row = (long long)ytab[yy] * Win;
        const int r = iters - (dy < 0 ? -dy : dy);
        for (int dx = -r; dx <= r; ++dx) {
            const int xx = x + dx;
            if (xx < 0 || xx >= W) continue;
            if (s[row + xtab[xx]] != 0) { hit = 1; break; }
        }
    }
    dst[idx] = (unsigned char)hit;
}

// masked, normalised, mirror-padded clip (test.py:146-165): out[ti][c][y][x] = (frame/255*2-1) * (1 - mask), rows / columns
// past the frame take the flipped image (cat([x, flip(x)])[: h + pad])
__global__ void masked_clip_kernel(const unsigned char* __restrict__ frames, const unsigned char* __restrict__ masks,
                                   const int* __restrict__ ids, float* __restrict__ out, int t, int H, int W, int Hp, int Wp) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)t * 3 * Hp * Wp) return;
    const int x = (int)(idx % Wp);
    long long r = idx / Wp;
    const int y = (int)(r % Hp);
    r /= Hp;
    const int c = (int)(r % 3);
    const int ti = (int)(r / 3);
    const int sy = y < H ? y : 2 * H - 1 - y;
    const int sx = x < W ? x : 2 * W - 1 - x;
    const long long pix = ((long long)ids[ti] * H + sy) * W + sx;
    const float v = ((float)frames[pix * 3 + c] / 255.0f) * 2.0f - 1.0f;
    const float m = (float)masks[pix];
    out[idx] = v * (1.0f - m);
}

// compositing + 0.5/0.5 blending of overlapping windows (test.py:168-179):
//   img = uint8((pred+1)/2*255) * mask + frame * (1-mask);  comp = img (first time)  or  comp*0.5 + img*0.5
__global__ void composite_kernel(const float* __restrict__ pred, const int* __restrict__ ids, const unsigned char* __restrict__ first,
                                 const unsigned char* __restrict__ frames, const unsigned char* __restrict__ masks,
                                 float* __restrict__ comp, int n, int H, int W, int Hp, int Wp) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * H * W * 3) return;
    const int c = (int)(idx % 3);
    long long r = idx / 3;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int i = (int)(r / H);
    const int f = ids[i];
    const long long pix = ((long long)f * H + y) * W + x;
    unsigned char img = frames[pix * 3 + c];
    if (masks[pix]) {
        const float p = pred[(((long long)i * 3 + c) * Hp + y) * Wp + x];
        img = (unsigned char)(int)(((p + 1.0f) / 2.0f) * 255.0f);
    }
    const float v = (float)img;
    comp[pix * 3 + c] = first[i] ? v : comp[pix * 3 + c] * 0.5f + v * 0.5f;
}

__global__ void float_to_u8_kernel(const float* __restrict__ src, unsigned char* __restrict__ dst, long long n) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n) dst[idx] = (unsigned char)(int)src[idx];
}

// model output NCHW float in (-1,1) -> NHWC uint8 (what test.py:168-171 turns a prediction into), cropped to H x W
__global__ void pred_to_u8_kernel(const float* __restrict__ pred, unsigned char* __restrict__ dst, int N, int H, int W, int Hp, int Wp) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * H * W * 3) return;
    const int c = (int)(idx % 3);
    long long r = idx / 3;
    const int x = (int)(r % W);
    r /= W;
    const int y = (int)(r % H);
    const int n = (int)(r / H);
    const float p = pred[(((long long)n * 3 + c) * Hp + y) * Wp + x];
    dst[idx] = (unsigned char)(int)(((p + 1.0f) / 2.0f) * 255.0f);
}

}  // namespace

extern "C" int e2fgvi_mask_prepare(const uint8_t* masks, int32_t L, int32_t Hin, int32_t Win, const int32_t* ytab,
                                   const int32_t* xtab, uint8_t* out, int32_t H, int32_t W, int32_t iterations, void* stream) {
    E2_REQUIRE(masks && ytab && xtab && out && L > 0 && Hin > 0 && Win > 0 && H > 0 && W > 0 && iterations >= 0 && iterations <= 64,
               E2FGVI_EINVAL, "mask_prepare: bad arguments");
    hipLaunchKernelGGL(mask_prepare_kernel, dim3(blocks_for((long long)L * H * W)), dim3(NTH), 0, (hipStream_t)stream, masks, Hin,
                       Win, ytab, xtab, out, L, H, W, iterations);
    E2_LAUNCH_CHECK("mask_prepare");
    return 0;
}

extern "C" int e2fgvi_masked_clip(const uint8_t* frames, const uint8_t* masks, const int32_t* ids, int32_t t, int32_t H, int32_t W,
                                  float* clip, int32_t Hp, int32_t Wp, void* stream) {
    E2_REQUIRE(frames && masks && ids && clip && t > 0 && H > 0 && W > 0, E2FGVI_EINVAL, "masked_clip: bad arguments");
    E2_REQUIRE(Hp >= H && Wp >= W && Hp <= 2 * H && Wp <= 2 * W, E2FGVI_EINVAL, "masked_clip: padded size must be in [size, 2 size]");
    hipLaunchKernelGGL(masked_clip_kernel, dim3(blocks_for((long long)t * 3 * Hp * Wp)), dim3(NTH), 0, (hipStream_t)stream, frames,
                       masks, ids, clip, t, H, W, Hp, Wp);
    E2_LAUNCH_CHECK("masked_clip");
    return 0;
}

extern "C" int e2fgvi_composite(const float* pred, const int32_t* ids, const uint8_t* first, int32_t n, const uint8_t* frames,
                                const uint8_t* masks, float* comp, int32_t H, int32_t W, int32_t Hp, int32_t Wp, void* stream) {
    E2_REQUIRE(pred && ids && first && frames && masks && comp && n > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W, E2FGVI_EINVAL,
               "composite: bad arguments");
    hipLaunchKernelGGL(composite_kernel, dim3(blocks_for((long long)n * H * W * 3)), dim3(NTH), 0, (hipStream_t)stream, pred, ids,
                       first, frames, masks, comp, n, H, W, Hp, Wp);
    E2_LAUNCH_CHECK("composite");
    return 0;
}

extern "C" int e2fgvi_float_to_u8(const float* src, uint8_t* dst, int64_t n, void* stream) {
    E2_REQUIRE(src && dst && n > 0, E2FGVI_EINVAL, "float_to_u8: bad arguments");
    hipLaunchKernelGGL(float_to_u8_kernel, dim3(blocks_for(n)), dim3(NTH), 0, (hipStream_t)stream, src, dst, (long long)n);
    E2_LAUNCH_CHECK("float_to_u8");
    return 0;
}

extern "C" int e2fgvi_pred_to_u8(const float* pred, uint8_t* dst, int32_t N, int32_t H, int32_t W, int32_t Hp, int32_t Wp,
                                 void* stream) {
    E2_REQUIRE(pred && dst && N > 0 && H > 0 && W > 0 && Hp >= H && Wp >= W, E2FGVI_EINVAL, "pred_to_u8: bad arguments");
    hipLaunchKernelGGL(pred_to_u8_kernel, dim3(blocks_for((long long)N * H * W * 3)), dim3(NTH), 0, (hipStream_t)stream, pred, dst, N,
                       H, W, Hp, Wp);
    E2_LAUNCH_CHECK("pred_to_u8");
    return 0;
}
