// HBM-bound helper kernels of the E2FGVI forward (gfx950): layout transforms, bilinear resizes,
// flow warps, LayerNorm, window pooling and the fold/unfold gathers.  All are pure data movement
// with a few flops per byte: the rules that matter are coalesced 16-byte accesses along the
// channel axis (NHWC) and no intermediate copies.  Reference call sites: include/e2fgvi_hip.h.
#include "common.h"

namespace {

constexpr int NTH = 256;

// 4 consecutive channels as fp32, from / to fp32 (16 bytes) or bf16 (8 bytes) tensors -- the bf16 data path
// (BASELINE.json configs 4 / 5) keeps its activations in HBM as bf16, every kernel here computes in fp32
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld4<__bf16>(const __bf16* p) {
    const u32x2 q = *reinterpret_cast<const u32x2*>(p);
    f32x4 v = {__builtin_bit_cast(float, q[0] << 16), __builtin_bit_cast(float, q[0] & 0xFFFF0000u),
               __builtin_bit_cast(float, q[1] << 16), __builtin_bit_cast(float, q[1] & 0xFFFF0000u)};
    return v;
}
// channels per thread: 4 for fp32 tensors, 8 for bf16 tensors -- a 16-byte access either way.  ldv / stv move VT<T>::N
// consecutive channels as VT<T>::Q fp32 quads; ldf loads the same channel count from an fp32 tensor.
typedef unsigned int u32x4m __attribute__((ext_vector_type(4)));
template <typename T> struct VT;
template <> struct VT<float> { static constexpr int N = 4, Q = 1; };
template <> struct VT<__bf16> { static constexpr int N = 8, Q = 2; };
template <typename T> __device__ __forceinline__ void ldv(const T* p, f32x4 (&v)[VT<T>::Q]);
template <> __device__ __forceinline__ void ldv<float>(const float* p, f32x4 (&v)[1]) { v[0] = *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ void ldv<__bf16>(const __bf16* p, f32x4 (&v)[2]) {
    const u32x4m q = *reinterpret_cast<const u32x4m*>(p);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        v[i][0] = __builtin_bit_cast(float, q[2 * i] << 16);     v[i][1] = __builtin_bit_cast(float, q[2 * i] & 0xFFFF0000u);
        v[i][2] = __builtin_bit_cast(float, q[2 * i + 1] << 16); v[i][3] = __builtin_bit_cast(float, q[2 * i + 1] & 0xFFFF0000u);
    }
}
template <typename T> __device__ __forceinline__ void stv(T* p, const f32x4 (&v)[VT<T>::Q]);
template <> __device__ __forceinline__ void stv<float>(float* p, const f32x4 (&v)[1]) { *reinterpret_cast<f32x4*>(p) = v[0]; }
template <> __device__ __forceinline__ void stv<__bf16>(__bf16* p, const f32x4 (&v)[2]) {
    typedef __bf16 bf16x8m __attribute__((ext_vector_type(8)));
    bf16x8m h = {(__bf16)v[0][0], (__bf16)v[0][1], (__bf16)v[0][2], (__bf16)v[0][3],
                 (__bf16)v[1][0], (__bf16)v[1][1], (__bf16)v[1][2], (__bf16)v[1][3]};
    *reinterpret_cast<bf16x8m*>(p) = h;
}
template <int Q> __device__ __forceinline__ void ldf(const float* p, f32x4 (&v)[Q]) {
#pragma unroll
    for (int i = 0; i < Q; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + 4 * i);
}
template <typename T> __device__ __forceinline__ void st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4<__bf16>(__bf16* p, f32x4 v) {
    bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
    *reinterpret_cast<bf16x4*>(p) = h;
}

// Flat thread index -> coordinates.  A 64-bit division by a run-time value is ~100 instructions on this ISA, a 32-bit
// unsigned one ~25: with three or four of them per thread the 64-bit form cost more than the memory traffic of these
// HBM-bound kernels (the x2 bilinear upsample of the decoder ran at 350 us per call at 720p against a 120-240 us floor).
// Every launch whose flat extent fits 32 bits -- all of them up to 1080p -- takes the 32-bit path (a uniform branch).
struct FlatIdx {
    unsigned u;
    long long l;
    bool small;
    __device__ __forceinline__ FlatIdx(long long idx, long long total) : u((unsigned)idx), l(idx), small(total <= 0xFFFFFFFFLL) {}
    // returns (index % d) and keeps (index / d)
    __device__ __forceinline__ int pop(int d) {
        if (small) {
            const unsigned q = u / (unsigned)d;
            const int r = (int)(u - q * (unsigned)d);
            u = q;
            return r;
        }
        const long long q = l / d;
        const int r = (int)(l - q * d);
        l = q;
        return r;
    }
    __device__ __forceinline__ long long rest() const { return small ? (long long)u : l; }
};

// ------------------------------------------------------------------------------------------ layout
template <typename TO>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, TO* __restrict__ dst, int C, int HW, int ld,
                                    float scale, float shift) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int c = c0 + r, p = p0 + threadIdx.x;
        tile[r][threadIdx.x] = (c < C && p < HW) ? src[((long long)n * C + c) * HW + p] * scale + shift : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int p = p0 + r, c = c0 + threadIdx.x;
        if (p < HW && c < ld) dst[((long long)n * HW + p) * ld + c] = (TO)((c < C) ? tile[threadIdx.x][r] : 0.f);
    }
}

// frames with few channels (C <= 8) -> NHWC pixels of exactly 8 channels (zeros beyond C): one thread per pixel, every
// plane read is coalesced across the lanes and the pixel leaves as one 16-byte (bf16) / two 16-byte (fp32) stores.  The
// tiled transpose above moves 32 B per 32-lane row for such tensors (358 us for ten 720x1296 frames; this: ~35 us).
template <typename TO>
__global__ void nchw_small_to_nhwc8_kernel(const float* __restrict__ src, TO* __restrict__ dst, int C, int HW, float scale,
                                           float shift, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    FlatIdx fi(idx, total);
    const int p = fi.pop(HW);
    const long long n = fi.rest();
    f32x4 v[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int c = 0; c < 8; ++c)
        if (c < C) v[c >> 2][c & 3] = src[(n * C + c) * HW + p] * scale + shift;
    st4(dst + idx * 8, v[0]);
    st4(dst + idx * 8 + 4, v[1]);
}

// ... and C <= 4 -> pixels of exactly 4 channels (the fp32 path's RGB + zero frames): one 16-byte (8-byte) store per pixel.  The
// tiled transpose spent 65 us on ten 240x432 frames at the head of the main stream (profiles/r05_timeline.txt).
template <typename TO>
__global__ void nchw_small_to_nhwc4_kernel(const float* __restrict__ src, TO* __restrict__ dst, int C, int HW, float scale,
                                           float shift, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    FlatIdx fi(idx, total);
    const int p = fi.pop(HW);
    const long long n = fi.rest();
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < C) v[c] = src[(n * C + c) * HW + p] * scale + shift;
    st4(dst + idx * 4, v);
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, int ld, float* __restrict__ dst, int C, int HW) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int p = p0 + r, c = c0 + threadIdx.x;
        tile[r][threadIdx.x] = (c < C && p < HW) ? src[((long long)n * HW + p) * ld + c] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int c = c0 + r, p = p0 + threadIdx.x;
        if (c < C && p < HW) dst[((long long)n * C + c) * HW + p] = tile[threadIdx.x][r];
    }
}

// ------------------------------------------------------------------------------------------ resize
// torch upsample_bilinear2d index rule
__device__ __forceinline__ void src_index(int d, float scale, int align, int in, int& i0, int& i1, float& l1) {
    float s = align ? scale * (float)d : fmaxf(scale * ((float)d + 0.5f) - 0.5f, 0.f);
    i0 = min((int)s, in - 1);
    i1 = i0 + ((i0 < in - 1) ? 1 : 0);
    l1 = s - (float)i0;
}

__global__ void resize_bilinear_kernel(const float* __restrict__ src, int src_nchw, int src_ld, float* __restrict__ dst,
                                       int dst_ld, int N, int C, int H, int W, int Ho, int Wo, int align, float sh,
                                       float sw, const float* __restrict__ scale, const float* __restrict__ shift,
                                       long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    FlatIdx fi(idx, total);
    const int c = fi.pop(C);
    const int ox = fi.pop(Wo);
    const int oy = fi.pop(Ho);
    const int n = (int)fi.rest();
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(oy, sh, align, H, y0, y1, ly);
    src_index(ox, sw, align, W, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    float v00, v01, v10, v11;
    if (src_nchw) {
        const float* b = src + ((long long)n * C + c) * H * W;
        v00 = b[y0 * W + x0]; v01 = b[y0 * W + x1]; v10 = b[y1 * W + x0]; v11 = b[y1 * W + x1];
    } else {
        const float* b = src + (long long)n * H * W * src_ld + c;
        v00 = b[((long long)y0 * W + x0) * src_ld]; v01 = b[((long long)y0 * W + x1) * src_ld];
        v10 = b[((long long)y1 * W + x0) * src_ld]; v11 = b[((long long)y1 * W + x1) * src_ld];
    }
    float v = hy * (hx * v00 + lx * v01) + ly * (hx * v10 + lx * v11);
    if (scale) v *= scale[c];
    if (shift) v += shift[c];
    dst[(((long long)n * Ho + oy) * Wo + ox) * dst_ld + c] = v;
}

// NHWC -> NHWC, 4 channels per thread (16-byte loads/stores): the decoder's x2 upsample moves 130-400 MB per call
template <typename T>
__global__ void resize_bilinear_vec4_kernel(const T* __restrict__ src, int src_ld, T* __restrict__ dst, int dst_ld,
                                            int N, int CV, int H, int W, int Ho, int Wo, int align, float sh, float sw,
                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                            long long total) {
    constexpr int NV = VT<T>::N, Q = VT<T>::Q;          // CV = C / NV channel vectors per pixel
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    FlatIdx fi(idx, total);
    const int cv = fi.pop(CV);
    const int ox = fi.pop(Wo);
    const int oy = fi.pop(Ho);
    const int n = (int)fi.rest();
    int y0, y1, x0, x1;
    float ly, lx;
    src_index(oy, sh, align, H, y0, y1, ly);
    src_index(ox, sw, align, W, x0, x1, lx);
    const float hy = 1.f - ly, hx = 1.f - lx;
    const T* b = src + (long long)n * H * W * src_ld + cv * NV;
    f32x4 v00[Q], v01[Q], v10[Q], v11[Q], v[Q];
    ldv(b + ((long long)y0 * W + x0) * src_ld, v00);
    ldv(b + ((long long)y0 * W + x1) * src_ld, v01);
    ldv(b + ((long long)y1 * W + x0) * src_ld, v10);
    ldv(b + ((long long)y1 * W + x1) * src_ld, v11);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
        v[q] = (v00[q] * hx + v01[q] * lx) * hy + (v10[q] * hx + v11[q] * lx) * ly;
        if (scale) v[q] = v[q] * *reinterpret_cast<const f32x4*>(scale + cv * NV + 4 * q);
        if (shift) v[q] = v[q] + *reinterpret_cast<const f32x4*>(shift + cv * NV + 4 * q);
    }
    stv(dst + (((long long)n * Ho + oy) * Wo + ox) * dst_ld + cv * NV, v);
}

__global__ void avgpool2_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int C, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int Ho = H / 2, Wo = W / 2;
    FlatIdx fi(idx, total);
    const int c = fi.pop(C);
    const int ox = fi.pop(Wo);
    const int oy = fi.pop(Ho);
    const long long n = fi.rest();
    const float* b = src + ((n * H + 2 * oy) * W + 2 * ox) * C + c;
    dst[idx] = (b[0] + b[C] + b[(long long)W * C] + b[(long long)W * C + C]) * 0.25f;
}

// ------------------------------------------------------------------------------------------ SPyNet
// one thread per (pair, y, x)
__global__ void spynet_level_input_kernel(const float* __restrict__ pyr, const int* __restrict__ ref_idx,
                                          const int* __restrict__ supp_idx, const float* __restrict__ flow_prev,
                                          float* __restrict__ out, __bf16* __restrict__ out16, int Np, int h, int w) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Np * h * w) return;
    FlatIdx fi(idx, (long long)Np * h * w);
    const int x = fi.pop(w);
    const int y = fi.pop(h);
    const int n = (int)fi.rest();
    float fu = 0.f, fv = 0.f;
    if (flow_prev) {   // flow_up = 2 * bilinear x2 (align_corners=True) of the previous level
        const int hp = h / 2, wp = w / 2;
        const float sh = hp > 1 ? (float)(hp - 1) / (float)(h - 1) : 0.f;
        const float sw = wp > 1 ? (float)(wp - 1) / (float)(w - 1) : 0.f;
        int y0, y1, x0, x1;
        float ly, lx;
        src_index(y, sh, 1, hp, y0, y1, ly);
        src_index(x, sw, 1, wp, x0, x1, lx);
        const float hy = 1.f - ly, hx = 1.f - lx;
        const float* b = flow_prev + (long long)n * hp * wp * 2;
        const float2 a00 = *reinterpret_cast<const float2*>(b + (y0 * wp + x0) * 2);
        const float2 a01 = *reinterpret_cast<const float2*>(b + (y0 * wp + x1) * 2);
        const float2 a10 = *reinterpret_cast<const float2*>(b + (y1 * wp + x0) * 2);
        const float2 a11 = *reinterpret_cast<const float2*>(b + (y1 * wp + x1) * 2);
        fu = 2.f * (hy * (hx * a00.x + lx * a01.x) + ly * (hx * a10.x + lx * a11.x));
        fv = 2.f * (hy * (hx * a00.y + lx * a01.y) + ly * (hx * a10.y + lx * a11.y));
    }
    const float* rimg = pyr + (long long)ref_idx[n] * h * w * 4;
    const float* simg = pyr + (long long)supp_idx[n] * h * w * 4;
    const f32x4 rv = *reinterpret_cast<const f32x4*>(rimg + ((long long)y * w + x) * 4);
    // border-padded bilinear warp (grid_sample padding_mode='border', align_corners=True)
    float px = fminf(fmaxf((float)x + fu, 0.f), (float)(w - 1));
    float py = fminf(fmaxf((float)y + fv, 0.f), (float)(h - 1));
    const float fx = floorf(px), fy = floorf(py);
    const int x0 = (int)fx, y0 = (int)fy;
    const int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
    const float lx = px - fx, ly = py - fy, hx = 1.f - lx, hy = 1.f - ly;
    const f32x4 s00 = *reinterpret_cast<const f32x4*>(simg + ((long long)y0 * w + x0) * 4);
    const f32x4 s01 = *reinterpret_cast<const f32x4*>(simg + ((long long)y0 * w + x1) * 4);
    const f32x4 s10 = *reinterpret_cast<const f32x4*>(simg + ((long long)y1 * w + x0) * 4);
    const f32x4 s11 = *reinterpret_cast<const f32x4*>(simg + ((long long)y1 * w + x1) * 4);
    const f32x4 sv = s00 * (hy * hx) + s01 * (hy * lx) + s10 * (ly * hx) + s11 * (ly * lx);
    float* o = out + idx * 8;
    f32x4 o0 = {rv[0], rv[1], rv[2], sv[0]};
    f32x4 o1 = {sv[1], sv[2], fu, fv};
    *reinterpret_cast<f32x4*>(o) = o0;
    *reinterpret_cast<f32x4*>(o + 4) = o1;
    if (out16) {                           // the same 8 channels as the bf16 source of the level's conv stack
        st4(out16 + idx * 8, o0);
        st4(out16 + idx * 8 + 4, o1);
    }
}

// ------------------------------------------------------------------------------------------ propagation
struct Bil {
    long long o00, o01, o10, o11;   // pixel offsets (in pixels) of the 4 corners, clamped
    float w00, w01, w10, w11;       // weights, zero for corners outside the image
};
__device__ __forceinline__ Bil bil_zeros(float px, float py, int H, int W) {
    Bil b;
    const float fx = floorf(px), fy = floorf(py);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float lx = px - fx, ly = py - fy, hx = 1.f - lx, hy = 1.f - ly;
    const bool vx0 = x0 >= 0 && x0 < W, vx1 = x1 >= 0 && x1 < W, vy0 = y0 >= 0 && y0 < H, vy1 = y1 >= 0 && y1 < H;
    // positions far outside (or NaN) give no valid corner
    const int cx0 = min(max(x0, 0), W - 1), cx1 = min(max(x1, 0), W - 1);
    const int cy0 = min(max(y0, 0), H - 1), cy1 = min(max(y1, 0), H - 1);
    b.o00 = (long long)cy0 * W + cx0; b.o01 = (long long)cy0 * W + cx1;
    b.o10 = (long long)cy1 * W + cx0; b.o11 = (long long)cy1 * W + cx1;
    b.w00 = (vy0 && vx0) ? hy * hx : 0.f;
    b.w01 = (vy0 && vx1) ? hy * lx : 0.f;
    b.w10 = (vy1 && vx0) ? ly * hx : 0.f;
    b.w11 = (vy1 && vx1) ? ly * lx : 0.f;
    return b;
}

// source loads of prop_cond: VT<TC>::N channels from an fp32 tensor, or (bf16 sources, bf16 cond) one 16-byte bf16 octet
template <typename TS, int Q> __device__ __forceinline__ void ld_src(const TS* p, f32x4 (&v)[Q]);
template <> __device__ __forceinline__ void ld_src<float, 1>(const float* p, f32x4 (&v)[1]) { ldf<1>(p, v); }
template <> __device__ __forceinline__ void ld_src<float, 2>(const float* p, f32x4 (&v)[2]) { ldf<2>(p, v); }
template <> __device__ __forceinline__ void ld_src<__bf16, 2>(const __bf16* p, f32x4 (&v)[2]) { ldv<__bf16>(p, v); }

// thread = (pixel, 4-channel chunk); C/4 threads per pixel
template <typename TC, typename TS>
__global__ void prop_cond_kernel(const TS* __restrict__ fp, int fp_ld, const TS* __restrict__ f2, int f2_ld,
                                 const float* __restrict__ flow_a, const float* __restrict__ flow_b,
                                 long long flow_img_stride, TC* __restrict__ cond, float* __restrict__ flows,
                                 __bf16* __restrict__ flows8, int N, int H, int W, int C) {
    constexpr int NV = VT<TC>::N, Q = VT<TC>::Q;
    const int cq = C / NV;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)N * H * W * cq) return;
    FlatIdx fi(idx, (long long)N * H * W * cq);
    const int c4 = fi.pop(cq);                         // channel vector index
    const long long pix = fi.rest();
    const int x = fi.pop(W);
    const int y = fi.pop(H);
    const int n = (int)fi.rest();
    const long long ip = (long long)y * W + x;
    const float* fa = flow_a + n * flow_img_stride;
    const float2 f1 = *reinterpret_cast<const float2*>(fa + ip * 2);
    const Bil b1 = bil_zeros((float)x + f1.x, (float)y + f1.y, H, W);
    float2 fl2 = make_float2(0.f, 0.f);
    f32x4 c1[Q], c2[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) c2[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (flow_b) {
        const float* fb = flow_b + n * flow_img_stride;
        const float2 a00 = *reinterpret_cast<const float2*>(fb + b1.o00 * 2);
        const float2 a01 = *reinterpret_cast<const float2*>(fb + b1.o01 * 2);
        const float2 a10 = *reinterpret_cast<const float2*>(fb + b1.o10 * 2);
        const float2 a11 = *reinterpret_cast<const float2*>(fb + b1.o11 * 2);
        fl2.x = f1.x + (a00.x * b1.w00 + a01.x * b1.w01 + a10.x * b1.w10 + a11.x * b1.w11);
        fl2.y = f1.y + (a00.y * b1.w00 + a01.y * b1.w01 + a10.y * b1.w10 + a11.y * b1.w11);
        const Bil b2 = bil_zeros((float)x + fl2.x, (float)y + fl2.y, H, W);
        const TS* s2 = f2 + (long long)n * H * W * f2_ld + c4 * NV;
        f32x4 a[Q], bq[Q], cc[Q], dd[Q];
        ld_src<TS, Q>(s2 + b2.o00 * f2_ld, a); ld_src<TS, Q>(s2 + b2.o01 * f2_ld, bq); ld_src<TS, Q>(s2 + b2.o10 * f2_ld, cc); ld_src<TS, Q>(s2 + b2.o11 * f2_ld, dd);
#pragma unroll
        for (int q = 0; q < Q; ++q) c2[q] = a[q] * b2.w00 + bq[q] * b2.w01 + cc[q] * b2.w10 + dd[q] * b2.w11;
    }
    {
        const TS* s1 = fp + (long long)n * H * W * fp_ld + c4 * NV;
        f32x4 a[Q], bq[Q], cc[Q], dd[Q];
        ld_src<TS, Q>(s1 + b1.o00 * fp_ld, a); ld_src<TS, Q>(s1 + b1.o01 * fp_ld, bq); ld_src<TS, Q>(s1 + b1.o10 * fp_ld, cc); ld_src<TS, Q>(s1 + b1.o11 * fp_ld, dd);
#pragma unroll
        for (int q = 0; q < Q; ++q) c1[q] = a[q] * b1.w00 + bq[q] * b1.w01 + cc[q] * b1.w10 + dd[q] * b1.w11;
    }
    TC* co = cond + pix * (2 * C);
    stv(co + c4 * NV, c1);
    stv(co + C + c4 * NV, c2);
    if (c4 == 0) {
        f32x4 fo = {f1.x, f1.y, fl2.x, fl2.y};
        *reinterpret_cast<f32x4*>(flows + pix * 4) = fo;
        if (flows8) {                      // the same four values as a bf16 conv source, padded to 8 channels
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            st4(flows8 + pix * 8, fo);
            st4(flows8 + pix * 8 + 4, z4);
        }
    }
}

// ------------------------------------------------------------------------------------------ LayerNorm
// one wave per row; C = 256 * VPL
template <int VPL, typename TO>
__global__ void layernorm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                 const float* __restrict__ beta, TO* __restrict__ y, long long rows, int C) {
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * C;
    f32x4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = *reinterpret_cast<const f32x4*>(xr + (i * 64 + lane) * 4);
        s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        v[i] = v[i] - mean;
        q += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.f / sqrtf(q / (float)C + 1e-5f);
    TO* yr = y + row * C;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + (i * 64 + lane) * 4);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(beta + (i * 64 + lane) * 4);
        st4(yr + (i * 64 + lane) * 4, v[i] * rstd * g + bb);
    }
}

// ------------------------------------------------------------------------------------------ window pooling
template <typename T>
__global__ void window_pool_kernel(const T* __restrict__ x, const float* __restrict__ w45,
                                   const float* __restrict__ bias1, T* __restrict__ pooled, int BT, int fh, int fw,
                                   int C) {
    constexpr int NV = VT<T>::N, Q = VT<T>::Q;
    const int cq = C / NV;
    const int nWw = fw / 9, nWh = fh / 5;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)BT * nWh * nWw * cq) return;
    FlatIdx fi(idx, (long long)BT * nWh * nWw * cq);
    const int c4 = fi.pop(cq);
    const int wx = fi.pop(nWw);
    const int wy = fi.pop(nWh);
    const long long bt = fi.rest();
    const float b = bias1[0];
    f32x4 acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = f32x4{b, b, b, b};
    for (int py = 0; py < 5; ++py)
        for (int px = 0; px < 9; ++px) {
            f32x4 v[Q];
            ldv(x + ((bt * fh + wy * 5 + py) * fw + wx * 9 + px) * C + c4 * NV, v);
#pragma unroll
            for (int q = 0; q < Q; ++q) acc[q] = acc[q] + v[q] * w45[py * 9 + px];
        }
    stv(pooled + idx * NV, acc);
}

// ------------------------------------------------------------------------------------------ fold / unfold (7,3,3)
// Patch channel order used by these kernels is [tap = ki*7+kj][c] (the Linear weights feeding /
// consuming them are permuted accordingly at pack time), so a tap's C channels are contiguous.
__device__ __forceinline__ void fold_range(int Y, int L, int& l_lo, int& l_hi) {
    // l such that 0 <= Y + 3 - 3l <= 6
    const int lo = Y - 3;                    // 3l >= Y-3
    l_lo = lo <= 0 ? 0 : (lo + 2) / 3;
    l_hi = min((Y + 3) / 3, L - 1);
}

// GELU, exact (erf) form of torch.nn.GELU (tfocal_transformer.py:82).  (Measured in round 2: Abramowitz-Stegun 7.1.26 on
// the hardware exp2 / rcp instead of libm's erff made this kernel SLOWER, 92 -> 114 us per call at 720p -- most hidden
// activations sit in erff's cheap polynomial branch.)
__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }

// GELU_OUT (with NORMALISE): the FFN's GELU applied to the folded value.  The reference applies it AFTER the unfold
// (tfocal_transformer.py:82,95-97), but the unfold is a pure gather with zero padding and GELU(0) = 0, so GELU commutes
// with it -- and the folded tensor has 5.4x fewer elements than the unfolded one (erff on 127 M values per block at 720p was
// the unfold kernel's bottleneck: 92 us against an HBM floor of 55).
template <bool NORMALISE, typename TE, typename TR, typename TD, bool GELU_OUT = false>
__global__ void fold_kernel(const TE* __restrict__ emb, const float* __restrict__ bias_hwc,
                            const TR* __restrict__ residual, TD* __restrict__ dst, int F, int fh, int fw, int H,
                            int W, int C) {
    static_assert(VT<TE>::N == VT<TD>::N && VT<TR>::N == VT<TD>::N, "fold: one element class per instantiation");
    constexpr int NV = VT<TD>::N, Q = VT<TD>::Q;
    const int cq = C / NV;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)F * H * W * cq) return;
    FlatIdx fi(idx, (long long)F * H * W * cq);
    const int c4 = fi.pop(cq);
    const int X = fi.pop(W);
    const int Y = fi.pop(H);
    const long long f = fi.rest();
    int ly0, ly1, lx0, lx1;
    fold_range(Y, fh, ly0, ly1);
    fold_range(X, fw, lx0, lx1);
    f32x4 acc[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int row_len = 49 * C;
    for (int ly = ly0; ly <= ly1; ++ly) {
        const int ki = Y + 3 - 3 * ly;
        for (int lx = lx0; lx <= lx1; ++lx) {
            const int kj = X + 3 - 3 * lx;
            f32x4 v[Q];
            ldv(emb + ((f * fh + ly) * fw + lx) * row_len + (ki * 7 + kj) * C + c4 * NV, v);
#pragma unroll
            for (int q = 0; q < Q; ++q) acc[q] = acc[q] + v[q];
        }
    }
    if (NORMALISE) {
        const float cnt = (float)((ly1 - ly0 + 1) * (lx1 - lx0 + 1));
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = acc[q] / cnt;
    }
    if (GELU_OUT) {
#pragma unroll
        for (int q = 0; q < Q; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[q][e] = gelu_exact(acc[q][e]);
    }
    const long long o = ((f * H + Y) * W + X) * C + c4 * NV;
    if (bias_hwc) {
        f32x4 bq[Q];
        ldf<Q>(bias_hwc + ((long long)Y * W + X) * C + c4 * NV, bq);
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = acc[q] + bq[q];
    }
    if (residual) {
        f32x4 rq[Q];
        ldv(residual + o, rq);
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[q] = acc[q] + rq[q];
    }
    stv(dst + o, acc);
}

template <typename T, bool GELU = true>
__global__ void unfold_gelu_kernel(const T* __restrict__ folded, T* __restrict__ out, int F, int fh, int fw,
                                   int H, int W, int C) {
    constexpr int NV = VT<T>::N, Q = VT<T>::Q;
    const int cq = C / NV;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)F * fh * fw * 49 * cq) return;
    FlatIdx fi(idx, (long long)F * fh * fw * 49 * cq);
    const int c4 = fi.pop(cq);
    const int tap = fi.pop(49);
    const int lx = fi.pop(fw);
    const int ly = fi.pop(fh);
    const long long f = fi.rest();
    const int Y = 3 * ly - 3 + tap / 7, X = 3 * lx - 3 + tap % 7;
    f32x4 v[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (Y >= 0 && Y < H && X >= 0 && X < W) {
        ldv(folded + ((f * H + Y) * W + X) * C + c4 * NV, v);
        if (GELU) {
#pragma unroll
            for (int q = 0; q < Q; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) v[q][e] = gelu_exact(v[q][e]);
        }
    }
    stv(out + idx * NV, v);
}

// fp32 <-> bf16 element conversion (4 elements per thread)
template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ src, TO* __restrict__ dst, long long n4) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < n4) st4(dst + idx * 4, ld4(src + idx * 4));
}

inline unsigned blocks_for(long long total) { return (unsigned)cdiv64(total, NTH); }

}  // namespace

#define E2_DT_OK(dt) ((dt) == E2FGVI_F32 || (dt) == E2FGVI_BF16)

extern "C" int e2fgvi_nchw_to_nhwc_x(const float* src, void* dst, int32_t dst_dtype, int32_t N, int32_t C, int32_t H, int32_t W,
                                     int32_t ld, float scale, float shift, void* stream) {
    E2_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && ld >= C && E2_DT_OK(dst_dtype), E2FGVI_EINVAL,
               "nchw_to_nhwc: bad arguments");
    if (ld == 8 && C <= 8 && ((uintptr_t)dst & 15) == 0) {
        const long long total = (long long)N * H * W;
        if (dst_dtype == E2FGVI_BF16)
            hipLaunchKernelGGL(nchw_small_to_nhwc8_kernel<__bf16>, dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream, src,
                               (__bf16*)dst, C, H * W, scale, shift, total);
        else
            hipLaunchKernelGGL(nchw_small_to_nhwc8_kernel<float>, dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream, src,
                               (float*)dst, C, H * W, scale, shift, total);
        E2_LAUNCH_CHECK("nchw_to_nhwc8");
        return 0;
    }
    if (ld == 4 && C <= 4 && ((uintptr_t)dst & 15) == 0) {
        const long long total = (long long)N * H * W;
        if (dst_dtype == E2FGVI_BF16)
            hipLaunchKernelGGL(nchw_small_to_nhwc4_kernel<__bf16>, dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream, src,
                               (__bf16*)dst, C, H * W, scale, shift, total);
        else
            hipLaunchKernelGGL(nchw_small_to_nhwc4_kernel<float>, dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream, src,
                               (float*)dst, C, H * W, scale, shift, total);
        E2_LAUNCH_CHECK("nchw_to_nhwc4");
        return 0;
    }
    dim3 grid(cdiv(H * W, 32), cdiv(ld, 32), N), block(32, 8);
    if (dst_dtype == E2FGVI_BF16)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<__bf16>, grid, block, 0, (hipStream_t)stream, src, (__bf16*)dst, C, H * W, ld, scale, shift);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, block, 0, (hipStream_t)stream, src, (float*)dst, C, H * W, ld, scale, shift);
    E2_LAUNCH_CHECK("nchw_to_nhwc");
    return 0;
}
extern "C" int e2fgvi_nchw_to_nhwc(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W, int32_t ld,
                                   float scale, float shift, void* stream) {
    return e2fgvi_nchw_to_nhwc_x(src, dst, E2FGVI_F32, N, C, H, W, ld, scale, shift, stream);
}

extern "C" int e2fgvi_nhwc_to_nchw(const float* src, int32_t ld, float* dst, int32_t N, int32_t C, int32_t H, int32_t W,
                                   void* stream) {
    E2_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && ld >= C, E2FGVI_EINVAL, "nhwc_to_nchw: bad arguments");
    dim3 grid(cdiv(H * W, 32), cdiv(C, 32), N), block(32, 8);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, block, 0, (hipStream_t)stream, src, ld, dst, C, H * W);
    E2_LAUNCH_CHECK("nhwc_to_nchw");
    return 0;
}

static void resize_scales(int H, int W, int Ho, int Wo, int align_corners, float& sh, float& sw) {
    if (align_corners) {
        sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
        sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    } else {
        sh = (float)H / (float)Ho;
        sw = (float)W / (float)Wo;
    }
}

extern "C" int e2fgvi_resize_bilinear(const float* src, int32_t src_nchw, int32_t src_ld, float* dst, int32_t dst_ld,
                                      int32_t N, int32_t C, int32_t H, int32_t W, int32_t Ho, int32_t Wo,
                                      int32_t align_corners, const float* scale, const float* shift, void* stream) {
    E2_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && dst_ld >= C &&
                   (src_nchw || src_ld >= C),
               E2FGVI_EINVAL, "resize_bilinear: bad arguments");
    float sh, sw;
    resize_scales(H, W, Ho, Wo, align_corners, sh, sw);
    const bool vec = !src_nchw && C % 4 == 0 && src_ld % 4 == 0 && dst_ld % 4 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0 &&
                     (!scale || ((uintptr_t)scale & 15) == 0) && (!shift || ((uintptr_t)shift & 15) == 0);
    if (vec) {
        const long long total4 = (long long)N * Ho * Wo * (C / 4);
        hipLaunchKernelGGL(resize_bilinear_vec4_kernel<float>, dim3(blocks_for(total4)), dim3(NTH), 0, (hipStream_t)stream, src,
                           src_ld, dst, dst_ld, N, C / 4, H, W, Ho, Wo, align_corners, sh, sw, scale, shift, total4);
        E2_LAUNCH_CHECK("resize_bilinear_vec4");
        return 0;
    }
    const long long total = (long long)N * Ho * Wo * C;
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream, src, src_nchw,
                       src_ld, dst, dst_ld, N, C, H, W, Ho, Wo, align_corners, sh, sw, scale, shift, total);
    E2_LAUNCH_CHECK("resize_bilinear");
    return 0;
}

extern "C" int e2fgvi_resize_bilinear_bf16(const void* src, int32_t src_ld, void* dst, int32_t dst_ld, int32_t N, int32_t C,
                                           int32_t H, int32_t W, int32_t Ho, int32_t Wo, int32_t align_corners, void* stream) {
    E2_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && dst_ld >= C && src_ld >= C && C % 8 == 0 &&
                   src_ld % 8 == 0 && dst_ld % 8 == 0 && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0,
               E2FGVI_EINVAL, "resize_bilinear_bf16: bad arguments (NHWC bf16, channels in multiples of 8)");
    float sh, sw;
    resize_scales(H, W, Ho, Wo, align_corners, sh, sw);
    const long long total4 = (long long)N * Ho * Wo * (C / 8);
    hipLaunchKernelGGL(resize_bilinear_vec4_kernel<__bf16>, dim3(blocks_for(total4)), dim3(NTH), 0, (hipStream_t)stream,
                       (const __bf16*)src, src_ld, (__bf16*)dst, dst_ld, N, C / 8, H, W, Ho, Wo, align_corners, sh, sw,
                       (const float*)nullptr, (const float*)nullptr, total4);
    E2_LAUNCH_CHECK("resize_bilinear_bf16");
    return 0;
}

extern "C" int e2fgvi_avgpool2_nhwc(const float* src, float* dst, int32_t N, int32_t H, int32_t W, int32_t C,
                                    void* stream) {
    E2_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, E2FGVI_EINVAL,
               "avgpool2: bad arguments");
    const long long total = (long long)N * (H / 2) * (W / 2) * C;
    hipLaunchKernelGGL(avgpool2_kernel, dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream, src, dst, H, W, C, total);
    E2_LAUNCH_CHECK("avgpool2");
    return 0;
}

extern "C" int e2fgvi_spynet_level_input_x(const float* pyr, const int32_t* ref_idx, const int32_t* supp_idx,
                                           const float* flow_prev, float* out, void* out_bf16, int32_t Np, int32_t h, int32_t w,
                                           void* stream) {
    E2_REQUIRE(pyr && ref_idx && supp_idx && out && Np > 0 && h > 0 && w > 0, E2FGVI_EINVAL, "spynet_level_input: bad arguments");
    E2_REQUIRE(!flow_prev || (h % 2 == 0 && w % 2 == 0), E2FGVI_EINVAL, "spynet_level_input: odd level size");
    const long long total = (long long)Np * h * w;
    hipLaunchKernelGGL(spynet_level_input_kernel, dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream, pyr, ref_idx,
                       supp_idx, flow_prev, out, (__bf16*)out_bf16, Np, h, w);
    E2_LAUNCH_CHECK("spynet_level_input");
    return 0;
}
extern "C" int e2fgvi_spynet_level_input(const float* pyr, const int32_t* ref_idx, const int32_t* supp_idx,
                                         const float* flow_prev, float* out, int32_t Np, int32_t h, int32_t w,
                                         void* stream) {
    return e2fgvi_spynet_level_input_x(pyr, ref_idx, supp_idx, flow_prev, out, nullptr, Np, h, w, stream);
}

extern "C" int e2fgvi_prop_cond_xs(const void* feat_prop, int32_t fp_ld, const void* feat_n2, int32_t f2_ld, int32_t src_dtype,
                                   const float* flow_a, const float* flow_b, int64_t flow_img_stride, void* cond,
                                   int32_t cond_dtype, float* flows, void* flows8_bf16, int32_t N, int32_t H, int32_t W,
                                   int32_t C, void* stream) {
    if (src_dtype == E2FGVI_F32)
        return e2fgvi_prop_cond_x((const float*)feat_prop, fp_ld, (const float*)feat_n2, f2_ld, flow_a, flow_b, flow_img_stride, cond,
                                  cond_dtype, flows, flows8_bf16, N, H, W, C, stream);
    E2_REQUIRE(src_dtype == E2FGVI_BF16 && cond_dtype == E2FGVI_BF16, E2FGVI_EINVAL, "prop_cond: bf16 sources need a bf16 cond");
    E2_REQUIRE(feat_prop && flow_a && cond && flows && N > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0 && fp_ld % 8 == 0 &&
                   ((uintptr_t)feat_prop & 15) == 0,
               E2FGVI_EINVAL, "prop_cond: bad arguments");
    E2_REQUIRE(!flow_b || (feat_n2 && f2_ld % 8 == 0 && ((uintptr_t)feat_n2 & 15) == 0), E2FGVI_EINVAL, "prop_cond: flow_b needs feat_n2");
    const long long total = (long long)N * H * W * (C / 8);
    hipLaunchKernelGGL((prop_cond_kernel<__bf16, __bf16>), dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream,
                       (const __bf16*)feat_prop, fp_ld, (const __bf16*)feat_n2, f2_ld, flow_a, flow_b, (long long)flow_img_stride,
                       (__bf16*)cond, flows, (__bf16*)flows8_bf16, N, H, W, C);
    E2_LAUNCH_CHECK("prop_cond");
    return 0;
}

extern "C" int e2fgvi_prop_cond_x(const float* feat_prop, int32_t fp_ld, const float* feat_n2, int32_t f2_ld,
                                  const float* flow_a, const float* flow_b, int64_t flow_img_stride, void* cond,
                                  int32_t cond_dtype, float* flows, void* flows8_bf16, int32_t N, int32_t H, int32_t W,
                                  int32_t C, void* stream) {
    E2_REQUIRE(feat_prop && flow_a && cond && flows && N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && fp_ld % 4 == 0 &&
                   E2_DT_OK(cond_dtype),
               E2FGVI_EINVAL, "prop_cond: bad arguments");
    E2_REQUIRE(!flow_b || (feat_n2 && f2_ld % 4 == 0), E2FGVI_EINVAL, "prop_cond: flow_b needs feat_n2");
    const int nv = cond_dtype == E2FGVI_BF16 ? 8 : 4;             // channels per thread
    E2_REQUIRE(C % nv == 0, E2FGVI_EINVAL, "prop_cond: C must be a multiple of %d", nv);
    const long long total = (long long)N * H * W * (C / nv);
    if (cond_dtype == E2FGVI_BF16)
        hipLaunchKernelGGL((prop_cond_kernel<__bf16, float>), dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream, feat_prop, fp_ld,
                           feat_n2, f2_ld, flow_a, flow_b, (long long)flow_img_stride, (__bf16*)cond, flows,
                           (__bf16*)flows8_bf16, N, H, W, C);
    else
        hipLaunchKernelGGL((prop_cond_kernel<float, float>), dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream, feat_prop, fp_ld,
                           feat_n2, f2_ld, flow_a, flow_b, (long long)flow_img_stride, (float*)cond, flows,
                           (__bf16*)flows8_bf16, N, H, W, C);
    E2_LAUNCH_CHECK("prop_cond");
    return 0;
}
extern "C" int e2fgvi_prop_cond(const float* feat_prop, int32_t fp_ld, const float* feat_n2, int32_t f2_ld,
                                const float* flow_a, const float* flow_b, int64_t flow_img_stride, float* cond,
                                float* flows, int32_t N, int32_t H, int32_t W, int32_t C, void* stream) {
    return e2fgvi_prop_cond_x(feat_prop, fp_ld, feat_n2, f2_ld, flow_a, flow_b, flow_img_stride, cond, E2FGVI_F32, flows,
                              nullptr, N, H, W, C, stream);
}

template <typename TO>
static int layernorm_launch(const float* x, const float* gamma, const float* beta, TO* y, int64_t rows, int32_t C, hipStream_t st) {
    const int wpb = 4;
    dim3 grid((unsigned)cdiv64(rows, wpb)), block(64 * wpb);
    switch (C / 256) {
        case 1: hipLaunchKernelGGL((layernorm_kernel<1, TO>), grid, block, 0, st, x, gamma, beta, y, (long long)rows, C); break;
        case 2: hipLaunchKernelGGL((layernorm_kernel<2, TO>), grid, block, 0, st, x, gamma, beta, y, (long long)rows, C); break;
        case 3: hipLaunchKernelGGL((layernorm_kernel<3, TO>), grid, block, 0, st, x, gamma, beta, y, (long long)rows, C); break;
        default: hipLaunchKernelGGL((layernorm_kernel<4, TO>), grid, block, 0, st, x, gamma, beta, y, (long long)rows, C); break;
    }
    E2_LAUNCH_CHECK("layernorm");
    return 0;
}
extern "C" int e2fgvi_layernorm_x(const float* x, const float* gamma, const float* beta, void* y, int32_t y_dtype, int64_t rows,
                                  int32_t C, void* stream) {
    E2_REQUIRE(x && gamma && beta && y && rows > 0 && E2_DT_OK(y_dtype), E2FGVI_EINVAL, "layernorm: bad arguments");
    E2_REQUIRE(C == 256 || C == 512 || C == 768 || C == 1024, E2FGVI_EUNSUP, "layernorm: C must be 256/512/768/1024");
    return y_dtype == E2FGVI_BF16 ? layernorm_launch(x, gamma, beta, (__bf16*)y, rows, C, (hipStream_t)stream)
                                  : layernorm_launch(x, gamma, beta, (float*)y, rows, C, (hipStream_t)stream);
}
extern "C" int e2fgvi_layernorm(const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int32_t C,
                                void* stream) {
    return e2fgvi_layernorm_x(x, gamma, beta, y, E2FGVI_F32, rows, C, stream);
}

extern "C" int e2fgvi_window_pool_x(const void* x, int32_t dtype, const float* w45, const float* bias1, void* pooled,
                                    int32_t BT, int32_t fh, int32_t fw, int32_t C, void* stream) {
    E2_REQUIRE(x && w45 && bias1 && pooled && BT > 0 && fh > 0 && fw > 0 && fh % 5 == 0 && fw % 9 == 0 && C % 4 == 0 &&
                   E2_DT_OK(dtype),
               E2FGVI_EINVAL, "window_pool: bad arguments");
    E2_REQUIRE(dtype == E2FGVI_F32 || C % 8 == 0, E2FGVI_EINVAL, "window_pool: bf16 needs C %% 8 == 0");
    const long long total = (long long)BT * (fh / 5) * (fw / 9) * (C / (dtype == E2FGVI_BF16 ? 8 : 4));
    if (dtype == E2FGVI_BF16)
        hipLaunchKernelGGL(window_pool_kernel<__bf16>, dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream,
                           (const __bf16*)x, w45, bias1, (__bf16*)pooled, BT, fh, fw, C);
    else
        hipLaunchKernelGGL(window_pool_kernel<float>, dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream,
                           (const float*)x, w45, bias1, (float*)pooled, BT, fh, fw, C);
    E2_LAUNCH_CHECK("window_pool");
    return 0;
}
extern "C" int e2fgvi_window_pool(const float* x, const float* w45, const float* bias1, float* pooled, int32_t BT,
                                  int32_t fh, int32_t fw, int32_t C, void* stream) {
    return e2fgvi_window_pool_x(x, E2FGVI_F32, w45, bias1, pooled, BT, fh, fw, C, stream);
}

static int check_fold(const char* name, int F, int fh, int fw, int H, int W, int C) {
    E2_REQUIRE(F > 0 && C > 0 && C % 4 == 0, E2FGVI_EINVAL, "%s: bad arguments", name);
    E2_REQUIRE(fh == (H + 6 - 7) / 3 + 1 && fw == (W + 6 - 7) / 3 + 1, E2FGVI_EINVAL,
               "%s: token grid %dx%d inconsistent with %dx%d", name, fh, fw, H, W);
    return 0;
}

extern "C" int e2fgvi_ffn_fold_x(const void* hid, void* folded, int32_t dtype, int32_t F, int32_t fh, int32_t fw, int32_t H,
                                 int32_t W, int32_t C, void* stream) {
    E2_REQUIRE(hid && folded && E2_DT_OK(dtype), E2FGVI_EINVAL, "ffn_fold: bad arguments");
    if (int rc = check_fold("ffn_fold", F, fh, fw, H, W, C)) return rc;
    E2_REQUIRE(dtype == E2FGVI_F32 || C % 8 == 0, E2FGVI_EINVAL, "ffn_fold: bf16 needs C %% 8 == 0");
    const long long total = (long long)F * H * W * (C / (dtype == E2FGVI_BF16 ? 8 : 4));
    if (dtype == E2FGVI_BF16)
        hipLaunchKernelGGL((fold_kernel<true, __bf16, __bf16, __bf16>), dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream,
                           (const __bf16*)hid, (const float*)nullptr, (const __bf16*)nullptr, (__bf16*)folded, F, fh, fw, H, W, C);
    else
        hipLaunchKernelGGL((fold_kernel<true, float, float, float>), dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream,
                           (const float*)hid, (const float*)nullptr, (const float*)nullptr, (float*)folded, F, fh, fw, H, W, C);
    E2_LAUNCH_CHECK("ffn_fold");
    return 0;
}
extern "C" int e2fgvi_ffn_fold(const float* hid, float* folded, int32_t F, int32_t fh, int32_t fw, int32_t H, int32_t W,
                               int32_t C, void* stream) {
    return e2fgvi_ffn_fold_x(hid, folded, E2FGVI_F32, F, fh, fw, H, W, C, stream);
}

extern "C" int e2fgvi_ffn_unfold_gelu_x(const void* folded, void* out, int32_t dtype, int32_t F, int32_t fh, int32_t fw,
                                        int32_t H, int32_t W, int32_t C, void* stream) {
    E2_REQUIRE(folded && out && E2_DT_OK(dtype), E2FGVI_EINVAL, "ffn_unfold_gelu: bad arguments");
    if (int rc = check_fold("ffn_unfold_gelu", F, fh, fw, H, W, C)) return rc;
    E2_REQUIRE(dtype == E2FGVI_F32 || C % 8 == 0, E2FGVI_EINVAL, "ffn_unfold_gelu: bf16 needs C %% 8 == 0");
    const long long total = (long long)F * fh * fw * 49 * (C / (dtype == E2FGVI_BF16 ? 8 : 4));
    if (dtype == E2FGVI_BF16)
        hipLaunchKernelGGL(unfold_gelu_kernel<__bf16>, dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream,
                           (const __bf16*)folded, (__bf16*)out, F, fh, fw, H, W, C);
    else
        hipLaunchKernelGGL(unfold_gelu_kernel<float>, dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream,
                           (const float*)folded, (float*)out, F, fh, fw, H, W, C);
    E2_LAUNCH_CHECK("ffn_unfold_gelu");
    return 0;
}
extern "C" int e2fgvi_ffn_unfold_gelu(const float* folded, float* out, int32_t F, int32_t fh, int32_t fw, int32_t H,
                                      int32_t W, int32_t C, void* stream) {
    return e2fgvi_ffn_unfold_gelu_x(folded, out, E2FGVI_F32, F, fh, fw, H, W, C, stream);
}

// The same pair with the GELU moved in front of the unfold (see fold_kernel): folded = GELU(fold(hid) / count), then a pure
// gather.  fp32: bit-identical to ffn_fold + ffn_unfold_gelu; bf16: one rounding less (GELU sees the unrounded fold).
extern "C" int e2fgvi_ffn_fold_gelu_x(const void* hid, void* folded, int32_t dtype, int32_t F, int32_t fh, int32_t fw, int32_t H,
                                      int32_t W, int32_t C, void* stream) {
    E2_REQUIRE(hid && folded && E2_DT_OK(dtype), E2FGVI_EINVAL, "ffn_fold_gelu: bad arguments");
    if (int rc = check_fold("ffn_fold_gelu", F, fh, fw, H, W, C)) return rc;
    E2_REQUIRE(dtype == E2FGVI_F32 || C % 8 == 0, E2FGVI_EINVAL, "ffn_fold_gelu: bf16 needs C %% 8 == 0");
    const long long total = (long long)F * H * W * (C / (dtype == E2FGVI_BF16 ? 8 : 4));
    if (dtype == E2FGVI_BF16)
        hipLaunchKernelGGL((fold_kernel<true, __bf16, __bf16, __bf16, true>), dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream,
                           (const __bf16*)hid, (const float*)nullptr, (const __bf16*)nullptr, (__bf16*)folded, F, fh, fw, H, W, C);
    else
        hipLaunchKernelGGL((fold_kernel<true, float, float, float, true>), dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream,
                           (const float*)hid, (const float*)nullptr, (const float*)nullptr, (float*)folded, F, fh, fw, H, W, C);
    E2_LAUNCH_CHECK("ffn_fold_gelu");
    return 0;
}
extern "C" int e2fgvi_ffn_unfold_x(const void* folded, void* out, int32_t dtype, int32_t F, int32_t fh, int32_t fw, int32_t H,
                                   int32_t W, int32_t C, void* stream) {
    E2_REQUIRE(folded && out && E2_DT_OK(dtype), E2FGVI_EINVAL, "ffn_unfold: bad arguments");
    if (int rc = check_fold("ffn_unfold", F, fh, fw, H, W, C)) return rc;
    E2_REQUIRE(dtype == E2FGVI_F32 || C % 8 == 0, E2FGVI_EINVAL, "ffn_unfold: bf16 needs C %% 8 == 0");
    const long long total = (long long)F * fh * fw * 49 * (C / (dtype == E2FGVI_BF16 ? 8 : 4));
    if (dtype == E2FGVI_BF16)
        hipLaunchKernelGGL((unfold_gelu_kernel<__bf16, false>), dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream,
                           (const __bf16*)folded, (__bf16*)out, F, fh, fw, H, W, C);
    else
        hipLaunchKernelGGL((unfold_gelu_kernel<float, false>), dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream,
                           (const float*)folded, (float*)out, F, fh, fw, H, W, C);
    E2_LAUNCH_CHECK("ffn_unfold");
    return 0;
}

extern "C" int e2fgvi_softcomp_fold(const float* emb, const float* bias_hwc, const float* residual, float* dst, int32_t F,
                                    int32_t fh, int32_t fw, int32_t H, int32_t W, int32_t C, void* stream) {
    E2_REQUIRE(emb && dst, E2FGVI_EINVAL, "softcomp_fold: null pointer");
    if (int rc = check_fold("softcomp_fold", F, fh, fw, H, W, C)) return rc;
    const long long total = (long long)F * H * W * (C / 4);
    hipLaunchKernelGGL((fold_kernel<false, float, float, float>), dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream, emb,
                       bias_hwc, residual, dst, F, fh, fw, H, W, C);
    E2_LAUNCH_CHECK("softcomp_fold");
    return 0;
}
/* bf16 data path: emb, residual and dst are bf16 (bias_hwc stays fp32) */
extern "C" int e2fgvi_softcomp_fold_bf16(const void* emb, const float* bias_hwc, const void* residual, void* dst, int32_t F,
                                         int32_t fh, int32_t fw, int32_t H, int32_t W, int32_t C, void* stream) {
    E2_REQUIRE(emb && dst, E2FGVI_EINVAL, "softcomp_fold_bf16: null pointer");
    if (int rc = check_fold("softcomp_fold_bf16", F, fh, fw, H, W, C)) return rc;
    E2_REQUIRE(C % 8 == 0, E2FGVI_EINVAL, "softcomp_fold_bf16: C %% 8 != 0");
    const long long total = (long long)F * H * W * (C / 8);
    hipLaunchKernelGGL((fold_kernel<false, __bf16, __bf16, __bf16>), dim3(blocks_for(total)), dim3(NTH), 0, (hipStream_t)stream,
                       (const __bf16*)emb, bias_hwc, (const __bf16*)residual, (__bf16*)dst, F, fh, fw, H, W, C);
    E2_LAUNCH_CHECK("softcomp_fold_bf16");
    return 0;
}

extern "C" int e2fgvi_cast(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t n, void* stream) {
    E2_REQUIRE(src && dst && n > 0 && n % 4 == 0 && E2_DT_OK(src_dtype) && E2_DT_OK(dst_dtype) && src_dtype != dst_dtype &&
                   (((uintptr_t)src | (uintptr_t)dst) & 7) == 0,
               E2FGVI_EINVAL, "cast: bad arguments (n must be a multiple of 4, dtypes must differ)");
    if (src_dtype == E2FGVI_F32)
        hipLaunchKernelGGL((cast_kernel<float, __bf16>), dim3(blocks_for(n / 4)), dim3(NTH), 0, (hipStream_t)stream,
                           (const float*)src, (__bf16*)dst, (long long)(n / 4));
    else
        hipLaunchKernelGGL((cast_kernel<__bf16, float>), dim3(blocks_for(n / 4)), dim3(NTH), 0, (hipStream_t)stream,
                           (const __bf16*)src, (float*)dst, (long long)(n / 4));
    E2_LAUNCH_CHECK("cast");
    return 0;
}

// ---- bf16 NHWC [P][C] -> [C / 16][P][16]: the planar source layout of the deformable conv (mdcn.hip, src_planar)
namespace {
__global__ void nhwc_to_planar16_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, long long P, int C8) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one 16-byte octet of 8 channels
    if (idx >= P * C8) return;
    const long long pix = idx / C8;
    const int o = (int)(idx - pix * C8);
    dst[((long long)(o >> 1) * P + pix) * 2 + (o & 1)] = src[idx];
}
}  // namespace

extern "C" int e2fgvi_nhwc_to_planar16(const void* src, void* dst, int64_t P, int32_t C, void* stream) {
    E2_REQUIRE(src && dst, E2FGVI_EINVAL, "nhwc_to_planar16: null pointer");
    E2_REQUIRE(P > 0 && C > 0 && C % 16 == 0 && ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, E2FGVI_EINVAL,
               "nhwc_to_planar16: C must be a multiple of 16, buffers 16-byte aligned");
    const long long n = (long long)P * (C / 8);
    hipLaunchKernelGGL(nhwc_to_planar16_kernel, dim3((unsigned)cdiv64(n, 256)), dim3(256), 0, (hipStream_t)stream,
                       (const uint4*)src, (uint4*)dst, (long long)P, C / 8);
    E2_LAUNCH_CHECK("nhwc_to_planar16");
    return 0;
}
