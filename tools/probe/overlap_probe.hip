// Minimal reproducer hunt for the round-1 cross-stream incident (DESIGN.md "Stream overlap"): `spynet_level_input` on one
// stream returned wrong warped values in lanes 48-63 of single waves while the bf16 conv tiles with 2x2 MFMA accumulators
// ran on another stream.  Standalone (no torch):
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/overlap_probe.hip -o /tmp/overlap_probe -ldl
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Ie2fgvi_amd/csrc -Iinclude tools/probe/conv_bf16_r1.hip \
//           e2fgvi_amd/csrc/error.hip -o /tmp/libe2fgvi_r1_aggressor.so          (round 1's bf16 tiles: the aggressor)
//     /tmp/overlap_probe e2fgvi_amd/csrc/libe2fgvi_hip.so [trials] [/tmp/libe2fgvi_r1_aggressor.so]
// Victims   V0 the product kernel's body   V1 + s_waitcnt vmcnt(0) before the arithmetic   V2 + vmcnt(0) and 32 idle cycles
//           V3 nontemporal (L1-bypassing) loads   V4 arithmetic without packed-f32 instructions (volatile scalar FMAs)
// Aggressors (other stream, long-running, >= 1 workgroup per CU)
//           A0 none   A1 product bf16 conv tile 5 (through the C ABI)   A2 product bf16 conv tile 2 (never triggered in r1)
//           A3 register-only v_mfma_f32_32x32x16_bf16 loop, 4 accumulators (no memory traffic)
//           A4 register-only v_mfma_f32_32x32x2_f32 loop   A5 VALU-only loop   A6 product fp32 conv (tile 0)
// Every victim launch is compared bit by bit with the same kernel's output on an idle GPU; mismatches are binned by lane.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../../include/e2fgvi_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void src_index(int d, float scale, int in, int& i0, int& i1, float& l) {
    float s = (float)d * scale;
    i0 = (int)s; if (i0 > in - 1) i0 = in - 1;
    i1 = i0 + 1 < in ? i0 + 1 : in - 1;
    l = s - (float)i0;
}

template <int MODE>
__global__ void victim(const float* __restrict__ pyr, const int* __restrict__ ref_idx, const int* __restrict__ supp_idx,
                       const float* __restrict__ flow_prev, float* __restrict__ out, int Np, int h, int w) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)Np * h * w) return;
    const int x = (int)(idx % w);
    const int y = (int)((idx / w) % h);
    const int n = (int)(idx / ((long long)w * h));
    float fu = 0.f, fv = 0.f;
    {
        const int hp = h / 2, wp = w / 2;
        const float sh = (float)(hp - 1) / (float)(h - 1), sw = (float)(wp - 1) / (float)(w - 1);
        int y0, y1, x0, x1; float ly, lx;
        src_index(y, sh, hp, y0, y1, ly);
        src_index(x, sw, wp, x0, x1, lx);
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
    float px = fminf(fmaxf((float)x + fu, 0.f), (float)(w - 1));
    float py = fminf(fmaxf((float)y + fv, 0.f), (float)(h - 1));
    const float fx = floorf(px), fy = floorf(py);
    const int x0 = (int)fx, y0 = (int)fy;
    const int x1 = min(x0 + 1, w - 1), y1 = min(y0 + 1, h - 1);
    const float lx = px - fx, ly = py - fy, hx = 1.f - lx, hy = 1.f - ly;
    const f32x4* p_r = reinterpret_cast<const f32x4*>(rimg + ((long long)y * w + x) * 4);
    const f32x4* p00 = reinterpret_cast<const f32x4*>(simg + ((long long)y0 * w + x0) * 4);
    const f32x4* p01 = reinterpret_cast<const f32x4*>(simg + ((long long)y0 * w + x1) * 4);
    const f32x4* p10 = reinterpret_cast<const f32x4*>(simg + ((long long)y1 * w + x0) * 4);
    const f32x4* p11 = reinterpret_cast<const f32x4*>(simg + ((long long)y1 * w + x1) * 4);
    f32x4 rv, s00, s01, s10, s11;
    if (MODE == 3) {
        rv = __builtin_nontemporal_load(p_r); s00 = __builtin_nontemporal_load(p00); s01 = __builtin_nontemporal_load(p01);
        s10 = __builtin_nontemporal_load(p10); s11 = __builtin_nontemporal_load(p11);
    } else {
        rv = *p_r; s00 = *p00; s01 = *p01; s10 = *p10; s11 = *p11;
    }
    if (MODE == 1 || MODE == 2) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(rv), "+v"(s00), "+v"(s01), "+v"(s10), "+v"(s11) :: "memory");
        if (MODE == 2) asm volatile("s_nop 7\n s_nop 7\n s_nop 7\n s_nop 7" : "+v"(s00), "+v"(s01), "+v"(s10), "+v"(s11));
    }
    f32x4 sv;
    if (MODE == 4) {
        const float w00 = hy * hx, w01 = hy * lx, w10 = ly * hx, w11 = ly * lx;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float a = s00[c] * w00;
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(s01[c]), "v"(w01));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(s10[c]), "v"(w10));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a) : "v"(s11[c]), "v"(w11));
            sv[c] = a;
        }
    } else {
        sv = s00 * (hy * hx) + s01 * (hy * lx) + s10 * (ly * hx) + s11 * (ly * lx);
    }
    float* o = out + idx * 8;
    f32x4 o0 = {rv[0], rv[1], rv[2], sv[0]};
    f32x4 o1 = {sv[1], sv[2], fu, fv};
    *reinterpret_cast<f32x4*>(o) = o0;
    *reinterpret_cast<f32x4*>(o + 4) = o1;
}

// ---- pure-ALU victims: no memory operand anywhere near the arithmetic.  Every lane iterates x <- x*a + b on two
// independent values, either with ONE packed instruction (v_pk_fma_f32) or with two scalar v_fma_f32, and the kernel
// itself compares the two lanes' results against the same recurrence done with integer-exact inputs at the end.
template <bool PACKED>
__global__ __launch_bounds__(256) void alu_victim(unsigned* bad_lanes, float* out, int iters) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const int lane = threadIdx.x & 63;
    f32x2 x = {1.0f + lane, 2.0f + lane};
    const f32x2 a = {0.999f, 1.001f}, b = {0.5f, -0.25f};
    for (int i = 0; i < iters; ++i) {
        if (PACKED) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
        else { asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[0]) : "v"(a[0]), "v"(b[0]));
               asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[1]) : "v"(a[1]), "v"(b[1])); }
    }
    out[(size_t)blockIdx.x * 256 * 2 + threadIdx.x * 2] = x[0];
    out[(size_t)blockIdx.x * 256 * 2 + threadIdx.x * 2 + 1] = x[1];
}

// ---- aggressor built up from the product tile's ingredients: 2x2 accumulators of v_mfma_f32_32x32x16_bf16 fed by
// ds_read_b128 from LDS every step (LEVEL 1), + v_cvt_pk_bf16_f32 / ds_write_b64 restaging (LEVEL 2), + buffer loads
// of fresh fp32 data every step (LEVEL 3)
template <int LEVEL>
__global__ __launch_bounds__(256) void mfma_lds_aggr(const float* __restrict__ src, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) __bf16 lds[2][128 * 72 + 8 * 128 * 8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, i = lane & 31, h = lane >> 5;
    for (int k = tid; k < 2 * (128 * 72 + 8 * 128 * 8); k += 256) (&lds[0][0])[k] = (__bf16)(0.001f * (k % 977));
    __syncthreads();
    f32x16 acc[2][2] = {};
    f32x4 st4 = {0.f, 0.f, 0.f, 0.f};
    const int wm = wave >> 1, wn = wave & 1;
    for (int it = 0; it < iters; ++it) {
        const __bf16* sA = lds[it & 1];
        const __bf16* sB = sA + 128 * 72;
        if (LEVEL >= 3) st4 = *reinterpret_cast<const f32x4*>(src + ((size_t)(blockIdx.x * 256 + tid) * 4 + (size_t)(it & 1023) * 262144) % (1 << 24));
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            bf16x8 a[2], b[2];
            for (int tm = 0; tm < 2; ++tm) a[tm] = *reinterpret_cast<const bf16x8*>(sA + ((wm * 2 + tm) * 32 + i) * 72 + st * 16 + h * 8);
            for (int tn = 0; tn < 2; ++tn) b[tn] = *reinterpret_cast<const bf16x8*>(sB + ((2 * st + h) * 128 + (wn * 2 + tn) * 32 + i) * 8);
            for (int tm = 0; tm < 2; ++tm)
                for (int tn = 0; tn < 2; ++tn) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
        }
        if (LEVEL >= 2) {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            const f32x4 v = st4 + (float)it * 1e-6f;
            bf16x4 hh = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
            *reinterpret_cast<bf16x4*>(const_cast<__bf16*>(lds[(it & 1) ^ 1]) + (tid >> 3) * 72 + (tid & 7) * 4) = hh;
        }
        __syncthreads();
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc[0][0][r] + acc[0][1][r] + acc[1][0][r] + acc[1][1][r];
    if (s == 123.456f) sink[0] = s;
}

// ---- register-only aggressors
__global__ __launch_bounds__(256) void spin_mfma_bf16(float* sink, int iters) {
    bf16x8 a, b;
    for (int k = 0; k < 8; ++k) { a[k] = (__bf16)(0.001f * (threadIdx.x + k)); b[k] = (__bf16)(0.002f * (threadIdx.x - k)); }
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 123.456f) sink[0] = s;
}
__global__ __launch_bounds__(256) void spin_mfma_f32(float* sink, int iters) {
    float a = 0.001f * threadIdx.x, b = 0.002f * threadIdx.x;
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, c3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    if (s == 123.456f) sink[0] = s;
}
__global__ __launch_bounds__(256) void spin_valu(float* sink, int iters) {
    float a = 0.001f * threadIdx.x, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int i = 0; i < iters; ++i) { a = a * b + c; c = c * b + d; d = d * b + a; b = b * 0.99999f + 1e-6f; }
    if (a + c + d == 123.456f) sink[0] = a;
}

typedef int (*conv_fn)(const e2fgvi_conv_desc*, void*);
typedef int64_t (*size_bf_fn)(int32_t, int32_t, int32_t, int32_t, int32_t, const int32_t*);
typedef int (*pack_bf_fn)(const float*, void*, int32_t, int32_t, int32_t, int32_t, int32_t, const int32_t*, void*);
typedef int (*convx_fn)(const e2fgvi_convx_desc*, void*);
typedef int64_t (*size_fn)(int32_t, int32_t, int32_t, int32_t, int32_t, const int32_t*, int32_t);
typedef int (*pack_fn)(const float*, float*, int32_t, int32_t, int32_t, int32_t, int32_t, const int32_t*, int32_t, void*);

static void fill(std::vector<float>& v, unsigned seed) {
    for (size_t i = 0; i < v.size(); ++i) { seed = seed * 1664525u + 1013904223u; v[i] = ((seed >> 8) & 0xFFFF) / 32768.f - 1.f; }
}

int main(int argc, char** argv) {
    const char* libpath = argc > 1 ? argv[1] : "e2fgvi_amd/csrc/libe2fgvi_hip.so";
    const int trials = argc > 2 ? atoi(argv[2]) : 60;
    void* L = dlopen(libpath, RTLD_NOW);
    if (!L) { printf("dlopen failed: %s\n", dlerror()); return 1; }
    const char* r1path = argc > 3 ? argv[3] : "/tmp/libe2fgvi_r1_aggressor.so";
    void* L1 = dlopen(r1path, RTLD_NOW);
    if (!L1) { printf("dlopen of the round-1 aggressor library failed: %s\n", dlerror()); return 1; }
    conv_fn conv_bf16 = (conv_fn)dlsym(L1, "e2fgvi_conv2d_nhwc_bf16");
    conv_fn conv_f32 = (conv_fn)dlsym(L, "e2fgvi_conv2d_nhwc");
    size_bf_fn size_bf = (size_bf_fn)dlsym(L1, "e2fgvi_packed_conv_weight_bf16_size");
    pack_bf_fn pack_bf = (pack_bf_fn)dlsym(L1, "e2fgvi_pack_conv_weight_bf16");
    size_fn size_f = (size_fn)dlsym(L, "e2fgvi_packed_conv_weight_size");
    pack_fn pack_f = (pack_fn)dlsym(L, "e2fgvi_pack_conv_weight");
    convx_fn conv_x = (convx_fn)dlsym(L, "e2fgvi_conv2d_bf16x");
    size_bf_fn size_x = (size_bf_fn)dlsym(L, "e2fgvi_packed_conv_weight_bf16x_size");
    pack_bf_fn pack_x = (pack_bf_fn)dlsym(L, "e2fgvi_pack_conv_weight_bf16x");
    hipStream_t sa, sb;
    CK(hipStreamCreate(&sa)); CK(hipStreamCreate(&sb));

    // ---- victim data: SPyNet level 5 of a 4-pair batch (64x128), previous-level flow of a few pixels
    const int Np = 4, F = 3, h = 64, w = 128;
    std::vector<float> hp((size_t)F * h * w * 4), hf((size_t)Np * (h / 2) * (w / 2) * 2);
    fill(hp, 1); fill(hf, 2);
    for (auto& v : hf) v *= 3.f;
    int hr[4] = {0, 1, 1, 2}, hs[4] = {1, 2, 0, 1};
    float *pyr, *flow, *vout, *vref; int *ridx, *sidx;
    const size_t out_n = (size_t)Np * h * w * 8;
    CK(hipMalloc(&pyr, hp.size() * 4)); CK(hipMalloc(&flow, hf.size() * 4)); CK(hipMalloc(&vout, out_n * 4)); CK(hipMalloc(&vref, out_n * 4));
    CK(hipMalloc(&ridx, 16)); CK(hipMalloc(&sidx, 16));
    CK(hipMemcpy(pyr, hp.data(), hp.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(flow, hf.data(), hf.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(ridx, hr, 16, hipMemcpyHostToDevice)); CK(hipMemcpy(sidx, hs, 16, hipMemcpyHostToDevice));

    // ---- product conv as aggressor: encoder layer 10 geometry (640 -> 512, 2 groups, 3x3) on 3 x 60x108
    const int cN = 3, cH = 60, cW = 108, cpg[2] = {128, 192}, groups = 2, Cout = 512;
    std::vector<float> hx0((size_t)cN * cH * cW * 256), hx1((size_t)cN * cH * cW * 384), hw((size_t)Cout * 320 * 9);
    fill(hx0, 3); fill(hx1, 4); fill(hw, 5);
    for (auto& v : hw) v *= 0.02f;
    float *x0, *x1, *wraw, *wp32, *cout; void* wpb;
    CK(hipMalloc(&x0, hx0.size() * 4)); CK(hipMalloc(&x1, hx1.size() * 4)); CK(hipMalloc(&wraw, hw.size() * 4));
    CK(hipMalloc(&cout, (size_t)cN * cH * cW * Cout * 4));
    CK(hipMemcpy(x0, hx0.data(), hx0.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(x1, hx1.data(), hx1.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(wraw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    const int64_t nb = size_bf(Cout, groups, 3, 3, 2, cpg), nf = size_f(Cout, groups, 3, 3, 2, cpg, 32);
    CK(hipMalloc(&wpb, nb * 2)); CK(hipMalloc(&wp32, nf * 4));
    if (pack_bf(wraw, wpb, Cout, groups, 3, 3, 2, cpg, 0) || pack_f(wraw, wp32, Cout, groups, 3, 3, 2, cpg, 32, 0)) { printf("pack failed\n"); return 1; }
    CK(hipDeviceSynchronize());
    e2fgvi_conv_desc d; memset(&d, 0, sizeof(d));
    d.src[0] = x0; d.src[1] = x1; d.src_ld[0] = 256; d.src_ld[1] = 384; d.src_cpg[0] = 128; d.src_cpg[1] = 192; d.nsrc = 2;
    d.N = cN; d.H = cH; d.W = cW; d.Ho = cH; d.Wo = cW; d.KH = 3; d.KW = 3; d.stride = 1; d.pad = 1; d.groups = groups; d.Cout = Cout; d.bk = 32;
    d.dst = cout; d.dst_ld = Cout; d.act = 2; d.slope = 0.2f;
    // the bf16-data-path kernel (bf16 sources; contents irrelevant: the fp32 buffers are re-read as bf16 with half the ld)
    e2fgvi_convx_desc dx; memset(&dx, 0, sizeof(dx));
    void* wpx = nullptr; float* coutx = nullptr;
    if (conv_x) {
        const int64_t nx = size_x(Cout, groups, 3, 3, 2, cpg);
        CK(hipMalloc(&wpx, nx * 2)); CK(hipMalloc(&coutx, (size_t)cN * cH * cW * Cout * 4));
        if (pack_x(wraw, wpx, Cout, groups, 3, 3, 2, cpg, 0)) { printf("pack_x failed\n"); return 1; }
        CK(hipDeviceSynchronize());
        dx.src[0] = x0; dx.src[1] = x1; dx.src_ld[0] = 256; dx.src_ld[1] = 384; dx.src_cpg[0] = 128; dx.src_cpg[1] = 192; dx.nsrc = 2;
        dx.N = cN; dx.H = cH; dx.W = cW; dx.Ho = cH; dx.Wo = cW; dx.KH = 3; dx.KW = 3; dx.stride = 1; dx.pad = 1; dx.groups = groups; dx.Cout = Cout;
        dx.wpacked = wpx; dx.dst = coutx; dx.dst_ld = Cout; dx.dst_dtype = 0; dx.act = 2; dx.slope = 0.2f;
    }
    float* sink; CK(hipMalloc(&sink, 4));
    float* big; CK(hipMalloc(&big, (size_t)(1 << 24) * 4 + 4096)); CK(hipMemset(big, 0, (size_t)(1 << 24) * 4 + 4096));

    auto launch_victim = [&](int mode, float* o, hipStream_t st) {
        const long long total = (long long)Np * h * w;
        dim3 g((unsigned)((total + 255) / 256)), b(256);
        switch (mode) {
            case 0: hipLaunchKernelGGL(victim<0>, g, b, 0, st, pyr, ridx, sidx, flow, o, Np, h, w); break;
            case 1: hipLaunchKernelGGL(victim<1>, g, b, 0, st, pyr, ridx, sidx, flow, o, Np, h, w); break;
            case 2: hipLaunchKernelGGL(victim<2>, g, b, 0, st, pyr, ridx, sidx, flow, o, Np, h, w); break;
            case 3: hipLaunchKernelGGL(victim<3>, g, b, 0, st, pyr, ridx, sidx, flow, o, Np, h, w); break;
            default: hipLaunchKernelGGL(victim<4>, g, b, 0, st, pyr, ridx, sidx, flow, o, Np, h, w); break;
        }
    };
    auto launch_aggr = [&](int a, hipStream_t st) {
        switch (a) {
            case 1: d.wpacked = (const float*)wpb; d.tile = 5; for (int k = 0; k < 6; ++k) conv_bf16(&d, st); break;
            case 2: d.wpacked = (const float*)wpb; d.tile = 2; for (int k = 0; k < 6; ++k) conv_bf16(&d, st); break;
            case 3: hipLaunchKernelGGL(spin_mfma_bf16, dim3(512), dim3(256), 0, st, sink, 60000); break;
            case 4: hipLaunchKernelGGL(spin_mfma_f32, dim3(512), dim3(256), 0, st, sink, 8000); break;
            case 5: hipLaunchKernelGGL(spin_valu, dim3(2048), dim3(256), 0, st, sink, 200000); break;
            case 6: d.wpacked = wp32; d.tile = 0; for (int k = 0; k < 2; ++k) conv_f32(&d, st); break;
            case 7: d.wpacked = (const float*)wpb; d.tile = 1; for (int k = 0; k < 6; ++k) conv_bf16(&d, st); break;
            case 8: d.wpacked = (const float*)wpb; d.tile = 6; for (int k = 0; k < 6; ++k) conv_bf16(&d, st); break;
            case 9: d.wpacked = (const float*)wpb; d.tile = 3; for (int k = 0; k < 6; ++k) conv_bf16(&d, st); break;
            case 10: hipLaunchKernelGGL(mfma_lds_aggr<1>, dim3(512), dim3(256), 0, st, big, sink, 6000); break;
            case 11: hipLaunchKernelGGL(mfma_lds_aggr<2>, dim3(512), dim3(256), 0, st, big, sink, 6000); break;
            case 13: if (conv_x) { dx.tile = 1; for (int k = 0; k < 8; ++k) conv_x(&dx, st); } break;
            case 14: if (conv_x) { dx.tile = 2; for (int k = 0; k < 8; ++k) conv_x(&dx, st); } break;
            case 12: hipLaunchKernelGGL(mfma_lds_aggr<3>, dim3(512), dim3(256), 0, st, big, sink, 6000); break;
            default: break;
        }
    };
    const char* an[] = {"none", "bf16 conv tile 5 (product)", "bf16 conv tile 2 (product)", "mfma bf16 32x32x16 spin (registers only)",
                        "mfma f32 32x32x2 spin (registers only)", "VALU spin", "fp32 conv (product)", "bf16 conv tile 1 (product)",
                        "bf16 conv tile 6 (product)", "bf16 conv tile 3 (product)", "2x2 mfma bf16 + ds_read_b128",
                        "2x2 mfma bf16 + ds_read + cvt_pk/ds_write", "2x2 mfma bf16 + ds_read + cvt/ds_write + loads",
                        "bf16x conv tile 1 (128x128, LDS-DMA)", "bf16x conv tile 2 (128x64, LDS-DMA)"};
    const int NA = 14;
    const char* vn[] = {"V0 product body", "V1 +vmcnt(0)", "V2 +vmcnt(0)+nops", "V3 nontemporal loads", "V4 no packed f32"};
    std::vector<float> got(out_n), ref(out_n);
    // how long do the aggressors run? (one timing each)
    for (int a = 1; a <= NA; ++a) {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        launch_aggr(a, sa); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, sa)); launch_aggr(a, sa); CK(hipEventRecord(e1, sa)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("aggressor A%d %-42s runs %.3f ms\n", a, an[a], ms);
    }
    const int modes = (argc > 3) ? atoi(argv[3]) : 5;
    for (int v = 0; v < modes; ++v) {
        launch_victim(v, vref, sb); CK(hipDeviceSynchronize());
        CK(hipMemcpy(ref.data(), vref, out_n * 4, hipMemcpyDeviceToHost));
        for (int a = 0; a <= NA; ++a) {
            long long bad_launch = 0, bad_elems = 0, lanes[64] = {0}, chan[8] = {0};
            int launches = 0;
            for (int t = 0; t < trials; ++t) {
                launch_aggr(a, sa);
                for (int k = 0; k < 8; ++k) {            // victims spread over the aggressor's run
                    launch_victim(v, vout, sb);
                    CK(hipStreamSynchronize(sb));
                    CK(hipMemcpy(got.data(), vout, out_n * 4, hipMemcpyDeviceToHost));
                    ++launches;
                    bool bad = false;
                    for (size_t i = 0; i < out_n; ++i)
                        if (memcmp(&got[i], &ref[i], 4)) { bad = true; ++bad_elems; ++lanes[(i / 8) & 63]; ++chan[i & 7]; }
                    bad_launch += bad;
                }
                CK(hipDeviceSynchronize());
            }
            printf("%-22s beside %-42s: %lld / %d launches wrong, %lld elements", vn[v], an[a], bad_launch, launches, bad_elems);
            if (bad_elems) {
                printf("  lanes:");
                for (int l = 0; l < 64; l += 16) { long long s = 0; for (int k = 0; k < 16; ++k) s += lanes[l + k]; printf(" [%d-%d]=%lld", l, l + 15, s); }
                printf("  out-channel:");
                for (int c = 0; c < 8; ++c) printf(" %lld", chan[c]);
            }
            printf("\n"); fflush(stdout);
        }
    }
    // ---- pure-ALU victims
    {
        const int blocks = 1024, iters = 4000;
        float *o1; unsigned* bl; CK(hipMalloc(&o1, (size_t)blocks * 512 * 4)); CK(hipMalloc(&bl, 256));
        std::vector<float> r0((size_t)blocks * 512), r1((size_t)blocks * 512);
        for (int packed = 1; packed >= 0; --packed) {
            if (packed) hipLaunchKernelGGL(alu_victim<true>, dim3(blocks), dim3(256), 0, sb, bl, o1, iters);
            else hipLaunchKernelGGL(alu_victim<false>, dim3(blocks), dim3(256), 0, sb, bl, o1, iters);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(r0.data(), o1, r0.size() * 4, hipMemcpyDeviceToHost));
            for (int a = 0; a <= NA; ++a) {
                long long bad_launch = 0, bad = 0, lanes[4] = {0};
                int launches = 0;
                for (int t = 0; t < trials; ++t) {
                    launch_aggr(a, sa);
                    for (int k = 0; k < 8; ++k) {
                        if (packed) hipLaunchKernelGGL(alu_victim<true>, dim3(blocks), dim3(256), 0, sb, bl, o1, iters);
                        else hipLaunchKernelGGL(alu_victim<false>, dim3(blocks), dim3(256), 0, sb, bl, o1, iters);
                        CK(hipStreamSynchronize(sb));
                        CK(hipMemcpy(r1.data(), o1, r1.size() * 4, hipMemcpyDeviceToHost));
                        ++launches;
                        bool b = false;
                        for (size_t i = 0; i < r1.size(); ++i) if (memcmp(&r0[i], &r1[i], 4)) { b = true; ++bad; ++lanes[((i / 2) & 63) >> 4]; }
                        bad_launch += b;
                    }
                    CK(hipDeviceSynchronize());
                }
                printf("ALU victim (%s) beside %-46s: %lld / %d launches wrong, %lld values, lanes [0-15]=%lld [16-31]=%lld [32-47]=%lld [48-63]=%lld\n",
                       packed ? "v_pk_fma_f32" : "2 x v_fma_f32", an[a], bad_launch, launches, bad, lanes[0], lanes[1], lanes[2], lanes[3]);
                fflush(stdout);
            }
        }
    }
    return 0;
}
