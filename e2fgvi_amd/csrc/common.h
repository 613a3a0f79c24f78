// Shared device/host helpers for libe2fgvi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/e2fgvi_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

void e2fgvi_set_error(const char* fmt, ...);

#define E2_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            e2fgvi_set_error(__VA_ARGS__);     \
            return (code);                     \
        }                                      \
    } while (0)

#define E2_LAUNCH_CHECK(name)                                                        \
    do {                                                                             \
        hipError_t e__ = hipGetLastError();                                          \
        if (e__ != hipSuccess) {                                                     \
            e2fgvi_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return (int)e__;                                                         \
        }                                                                            \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }

// ---------------------------------------------------------------------------------------------
// fp32 MFMA tile product shared by the conv and the deformable-conv kernels.
//
// One wave owns a (TM*32) x (TN*32) block of the output tile.  v_mfma_f32_32x32x2_f32 operand
// maps (cdna guide section 3): lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
// lane l holds D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31] in register r.
//
// The K order inside a tile is free as long as A and B agree, so each half-wave (h = l>>5) takes
// a run of 4 consecutive k (one ds_read_b128) and feeds 4 consecutive MFMAs from it:
//   step (m8, e): lanes of half h use k = 8*m8 + 4*h + e.
// LDS images:  A: [BM rows][LDA = BK+4 floats]  (row padding makes the b128 reads conflict-free)
//              B: [BK/4][BN][4]                (k-quad interleaved, lane-linear b128 reads)
// ---------------------------------------------------------------------------------------------
template <int TM, int TN, int BK, int LDA, int BN>
__device__ __forceinline__ void mma_ktile(const float* __restrict__ sA, const float* __restrict__ sB,
                                          f32x16 (&acc)[TM][TN], int a_row0, int b_col0, int lane) {
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int m8 = 0; m8 < BK / 8; ++m8) {
        f32x4 a[TM], b[TN];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
            a[tm] = *reinterpret_cast<const f32x4*>(sA + (a_row0 + tm * 32 + i) * LDA + (2 * m8 + h) * 4);
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
            b[tn] = *reinterpret_cast<const f32x4*>(sB + ((2 * m8 + h) * BN + b_col0 + tn * 32 + i) * 4);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][e], b[tn][e], acc[tm][tn], 0, 0, 0);
    }
}

// XCD-aware bijective block remap (8 XCDs, block b is dispatched to XCD b % 8): gives every XCD a
// contiguous run of logical tiles so neighbouring tiles share one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7;
    const int xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// sigmoid / tanh on the hardware exp2 + rcp (1 ulp each): absolute error <= ~5e-7 -- used by the DCN offset / mask
// post-processing epilogues (10 * tanh -> 5e-6 px), where libm's branchy tanhf / expf cost a few microseconds per launch
__device__ __forceinline__ float e2_fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float e2_fast_tanh(float x) {
    // 1 - 2 / (1 + e^{2x}): saturates correctly for large |x| (exp2 -> inf / 0)
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// ---------------------------------------------------------------------------------------------
// Exact three-way bf16 split of fp32 values, two at a time: x = hi + mid + lo in exact arithmetic, every piece a bf16 -- what the
// split-operand ("x3") kernels feed to v_mfma_f32_32x32x16_bf16.  H / M / L hold the pieces of x0 in their low and
// of x1 in their high half.  hi = x with its low 16 bits cleared, mid = the same of the exact remainder, lo = the rest (<= 8
// significant bits): and, sub, and, sub per value + three v_perm per pair = 11 full-rate VALU instructions per pair.
// Measured and rejected (round 5, profiles/r05_split_rne_ab.txt): a round-to-nearest split on v_cvt_pk_bf16_f32 + v_dot2c_f32_bf16
// (the dot product with the constant pair (-1, 0) / (0, -1) expands one half of a packed pair and subtracts it in one
// instruction: 7 instructions per pair, exact on the chip per tools/probe/split_probe.hip) is SLOWER -- the wide-tile Winograd
// kernel +4 ... +11 %, everything else within +-4 % -- the two instructions do not issue at the full VALU rate.  (The probe also
// found that hipcc folds the packed constant (-1, 0) into the inline constant -1.0, which the instruction reads as (0, -1).)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void e2_split2(float x0, float x1, unsigned& H, unsigned& M, unsigned& L) {
    const unsigned b0 = __builtin_bit_cast(unsigned, x0), b1 = __builtin_bit_cast(unsigned, x1);
    const float r0 = x0 - __builtin_bit_cast(float, b0 & 0xFFFF0000u), r1 = x1 - __builtin_bit_cast(float, b1 & 0xFFFF0000u);
    const unsigned rb0 = __builtin_bit_cast(unsigned, r0), rb1 = __builtin_bit_cast(unsigned, r1);
    const float s0 = r0 - __builtin_bit_cast(float, rb0 & 0xFFFF0000u), s1 = r1 - __builtin_bit_cast(float, rb1 & 0xFFFF0000u);
    H = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    M = __builtin_amdgcn_perm(rb1, rb0, 0x07060302u);
    L = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
}
// eight values (two channel quads) -> three operand fragments of eight bf16 each, element j of the pair in position j
typedef __bf16 e2_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int e2_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void e2_split8(const f32x4& v0, const f32x4& v1, e2_bf16x8& hi, e2_bf16x8& mid, e2_bf16x8& lo) {
    // (plain scalars first: __builtin_bit_cast of a vector-ELEMENT lvalue reads element 0 for every index with this hipcc)
    float x[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { x[j] = v0[j]; x[4 + j] = v1[j]; }
    e2_u32x4 H, M, L;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned h, m, l;
        e2_split2(x[2 * j], x[2 * j + 1], h, m, l);
        H[j] = h; M[j] = m; L[j] = l;
    }
    hi = __builtin_bit_cast(e2_bf16x8, H);
    mid = __builtin_bit_cast(e2_bf16x8, M);
    lo = __builtin_bit_cast(e2_bf16x8, L);
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    if (act == E2FGVI_ACT_RELU) return fmaxf(v, 0.f);
    if (act == E2FGVI_ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == E2FGVI_ACT_TANH) return tanhf(v);
    return v;
}
