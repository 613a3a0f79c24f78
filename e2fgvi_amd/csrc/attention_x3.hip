// Fused temporal focal window attention of the fp32 path on the bf16 matrix pipe (gfx950, round 3): the operator of
// attention.hip (tfocal_transformer.py:226-396, tfocal_transformer_hq.py:231-425) with fp32 inputs, fp32 softmax and fp32
// output, both matrix products issued as SIX v_mfma_f32_32x32x16_bf16 terms of exactly split operands -- the scheme of
// conv_bf16x.hip MODE 2: an fp32 number is the exact sum of three bf16 numbers (8 + 8 + 8 significand bits), every partial
// product is exact in the MFMA's fp32 accumulator, the three smallest of the nine are dropped (< 2^-22 of the product).
//   K, V  split ONCE per forward by e2fgvi_split3_kv (one HBM-bound pass over the k / v columns of the qkv rows, pooled rows
//         included) into three bf16 planes [3][rows][1024]; the tiles of all three planes go global -> LDS by LDS-DMA into the
//         layouts of attention_bf16.hip's round-3 kernel (K: chunk c of key k in slot c ^ (k & 15); V: row-major, slots
//         permuted for ds_read_b64_tr_b16)
//   Q     read as fp32, split in registers once per wave (3 x 32 VGPRs for the wave's 32 queries)
//   P     the softmax numerators, split in registers per tile (two 8-key fragments per lane)
//   S^T = K . Q^T and O^T += V^T . P exactly as in attention_bf16.hip (each lane holds 16 keys of one query; the PV product's
//   k index enumerates the keys in the order the lane already holds them).
// Built without packed-fp32 VALU (e2fgvi_amd/build.py): LDS-fed bf16 MFMA waves.
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int HD = 128, NH = 4, CQ = 1536, CP = 1024;      // CP: k + v columns of a plane row
constexpr int WS0 = 5, WS1 = 9, WTOK = 45, SLOTS = 210;
constexpr int TK = 32;
constexpr float LOG2E = 1.4426950408889634f;
constexpr int X_KB = TK * HD * 2;                 // 8 KB: one plane of a K tile, [32 keys][16 slots of 16 bytes]
constexpr int X_STAGE = 6 * X_KB;                 // K planes hi, mid, lo, then V planes hi, mid, lo

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void x_lds_void;
typedef int x_i32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4 x_lds_s16x4;

// one LDS-DMA piece (attention_bf16.hip::v2_dma16)
__device__ __forceinline__ void x_dma16(x_i32x4 rsrc, unsigned lds_dst, unsigned voff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ x_i32x4 x_rsrc_words(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    x_i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xFFFFu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

// x = hi + mid + lo with bf16 pieces (common.h, e2_split8)
__device__ __forceinline__ void x_split8(const f32x4& v0, const f32x4& v1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    e2_split8(v0, v1, hi, mid, lo);
}

// k / v columns (512 .. 1535) of `rows` fp32 qkv rows -> three bf16 planes [3][rows][1024]; one thread per 8 columns
__global__ void split3_kv_kernel(const float* __restrict__ src, unsigned short* __restrict__ planes, long long rows) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= rows * (CP / 8)) return;
    const long long row = idx / (CP / 8);
    const int c8 = (int)(idx - row * (CP / 8)) * 8;
    const float* s = src + row * CQ + 512 + c8;
    bf16x8 hi, mid, lo;
    x_split8(*reinterpret_cast<const f32x4*>(s), *reinterpret_cast<const f32x4*>(s + 4), hi, mid, lo);
    const long long plane = rows * CP;
    unsigned short* o = planes + row * CP + c8;
    *reinterpret_cast<bf16x8*>(o) = hi;
    *reinterpret_cast<bf16x8*>(o + plane) = mid;
    *reinterpret_cast<bf16x8*>(o + 2 * plane) = lo;
}

// KS = 2: two key groups of NW waves per workgroup.  Group kg walks the tiles kg, kg + 2, ... with its own LDS ring and the
// SAME queries; the partial results are merged at the end (flash-decoding inside a workgroup).  At the 432x240 shape there are
// only ~900 blocks of 32 queries for 1024 SIMDs: with one key group every SIMD holds ONE wave whose MFMA, softmax and LDS
// phases cannot overlap; two groups put two waves of half the length on a SIMD.  A ring is then K double-buffered + V
// single-buffered (72 KB; two rings fit the LDS): K(t + 1) is issued at the top of tile t, V(t + 1) after tile t's PV products
// (two barriers per tile).
template <int NW, int KS>
__global__ __launch_bounds__(64 * NW * KS, 1) void focal_attn_x3_kernel(const float* __restrict__ qkv, const int* __restrict__ key_tab,
                                                                  int tab_ld, const int* __restrict__ nkeys,
                                                                  float* __restrict__ out, int B, int T, int fh, int fw,
                                                                  const char* planes, unsigned planes_bytes, unsigned plane_stride,
                                                                  unsigned pooled_row0) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NT = 64 * NW;
    constexpr int PIECES = 8 / NW;                 // 1-KiB DMA pieces of one plane of a K (and of a V) tile per wave
    constexpr unsigned OOB = 0xFFFFF000u;          // a key row past the end: out of range in every plane (launcher: 3 planes < 0xFFFFF000)
    static_assert(NW == 2 || NW == 4 || NW == 8, "waves per workgroup");
    constexpr int RING = KS == 1 ? 2 * X_STAGE : 9 * X_KB;          // KS = 2: K stage 0, K stage 1, V (three planes each)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[KS * RING];
    __shared__ int stab[256];
    extern __shared__ __attribute__((aligned(16))) unsigned ktab[];       // byte offset of every key row in plane 0, OOB past the end

    const int kg = (KS == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x / NT));
    const int tid = threadIdx.x - kg * NT;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int nWw = fw / WS1, nWh = fh / WS0, nWin = nWh * nWw;
    const int nqc = (T * WTOK + 32 * NW - 1) / (32 * NW);
    const int logical = (int)blockIdx.x;
    const int qchunk = logical % nqc;
    const int wh = (logical / nqc) % (nWin * NH);
    const int b = logical / (nqc * nWin * NH);
    const int win = wh / NH, head = wh - win * NH;
    const int wy = win / nWw, wx = win - wy * nWw;
    const int NQ = T * WTOK;
    const int ntok = fh * fw;
    const x_i32x4 rsrc = x_rsrc_words(planes, planes_bytes);

    const int nv = nkeys[win];
    const int NK = T * nv;
    const int ntiles = (NK + TK - 1) / TK;
    const int nIter = (ntiles + KS - 1) / KS;       // tiles per key group (the last group may get an all-masked one)
    const int* tab = key_tab + (long long)win * tab_ld;
    for (int e = threadIdx.x; e < nv && e < 256; e += KS * NT) stab[e] = tab[e];
    __syncthreads();
    {   // the key-row table: entry k = (frame t = k / nv, slot s = k % nv), walked without divisions; KS * nIter tiles
        int t = 0, sl = threadIdx.x;
        while (sl >= nv) { sl -= nv; ++t; }
        const unsigned head_off = (unsigned)(head * HD * 2);
        for (int k = threadIdx.x; k < KS * nIter * TK; k += KS * NT) {
            unsigned e = OOB;
            if (k < NK) {
                const int ref = stab[sl];
                const bool pooled = ref < 0;
                const unsigned rowi = pooled ? pooled_row0 + (unsigned)((b * T + t) * nWin + (-(ref + 1))) : (unsigned)((b * T + t) * ntok + ref);
                e = rowi * (unsigned)(CP * 2) + head_off;
            }
            ktab[k] = e;
            sl += KS * NT;
            while (sl >= nv) { sl -= nv; ++t; }
        }
    }
    __syncthreads();

    // ---- this wave's 32 queries: 128 d as 8 operand octets per k-step, three planes
    const int q0 = (qchunk * NW + wave) * 32;
    const bool wave_active = q0 < NQ;
    auto query_row = [&](bool& ok) -> long long {
        const int qi = q0 + i;
        ok = qi < NQ;
        const int qq = ok ? qi : 0;
        const int t = qq / WTOK, pp = qq - t * WTOK;
        const int py = pp / WS1, px = pp - py * WS1;
        return (long long)(b * T + t) * ntok + (wy * WS0 + py) * fw + (wx * WS1 + px);
    };
    bf16x8 qh[8], qm[8], ql[8];
    {
        bool ok;
        const long long row = query_row(ok);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            f32x4 v0 = {0.f, 0.f, 0.f, 0.f}, v1 = v0;
            if (ok) {
                const float* qp = qkv + row * CQ + head * HD + kk * 16 + h * 8;
                v0 = *reinterpret_cast<const f32x4*>(qp);
                v1 = *reinterpret_cast<const f32x4*>(qp + 4);
            }
            x_split8(v0, v1, qh[kk], qm[kk], ql[kk]);
        }
    }
    const float qscale = 0.08838834764831845f * LOG2E;       // 128^-0.5 * log2(e)

    f32x16 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    // ---- DMA bookkeeping (per plane): piece pc = wave + NW * jp covers key rows 4 pc .. 4 pc + 3 of a tile; lane -> (row, slot)
    int d_row[PIECES];
    unsigned d_kb[PIECES], d_vb[PIECES];
#pragma unroll
    for (int jp = 0; jp < PIECES; ++jp) {
        const int pc = wave + NW * jp;
        const int row = 4 * pc + (lane >> 4), slot = lane & 15;
        d_row[jp] = row;
        d_kb[jp] = (unsigned)((slot ^ (row & 15)) * 16);
        d_vb[jp] = 1024u + (unsigned)(((((slot >> 2) ^ (row & 3)) << 2) | (slot & 3)) * 16);     // V sits 512 bf16 behind K
    }
    const unsigned smem_lds = (unsigned)(unsigned long long)(x_lds_void*)smem + (unsigned)(kg * RING);
    // KS = 2: the K planes of tile kt into K stage `kb`, or its V planes into the ring's V area
    auto issue_half = [&](int kt, unsigned lds_off, bool v_half) {
        const unsigned sk = smem_lds + lds_off + (unsigned)(wave * 1024);
        unsigned e[PIECES];
#pragma unroll
        for (int jp = 0; jp < PIECES; ++jp) e[jp] = ktab[kt * TK + d_row[jp]];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int jp = 0; jp < PIECES; ++jp) {
                const unsigned base = e[jp] == OOB ? OOB : e[jp] + (unsigned)pl * plane_stride;
                x_dma16(rsrc, __builtin_amdgcn_readfirstlane(sk + (unsigned)(pl * X_KB + jp * NW * 1024)),
                        base == OOB ? OOB : base + (v_half ? d_vb[jp] : d_kb[jp]));
            }
    };
    auto issue_tile = [&](int kt, int stage) {
        const unsigned sk = smem_lds + (unsigned)(stage * X_STAGE + wave * 1024);
        unsigned e[PIECES];
#pragma unroll
        for (int jp = 0; jp < PIECES; ++jp) e[jp] = ktab[kt * TK + d_row[jp]];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
#pragma unroll
            for (int jp = 0; jp < PIECES; ++jp) {
                // (an OOB entry + plane offset wraps for pl > 0: keep it out of range explicitly)
                const unsigned base = e[jp] == OOB ? OOB : e[jp] + (unsigned)pl * plane_stride;
                x_dma16(rsrc, __builtin_amdgcn_readfirstlane(sk + (unsigned)(pl * X_KB + jp * NW * 1024)), base == OOB ? OOB : base + d_kb[jp]);
                x_dma16(rsrc, __builtin_amdgcn_readfirstlane(sk + (unsigned)((3 + pl) * X_KB + jp * NW * 1024)), base == OOB ? OOB : base + d_vb[jp]);
            }
    };

    // ---- operand read addresses inside a plane (attention_bf16.hip)
    const int lam = lane & 15, kap = lam >> 2;
    const int key_l = 4 * h + kap;
    const int chunk_lo = 2 * ((i >> 4) & 1) + ((lam & 3) >> 1);
    const int v_base = key_l * 256 + ((kap << 2) | chunk_lo) * 16 + (lam & 1) * 8;
    const int k_base = i * 256 + ((h ^ (i & 15)) << 4);

#pragma unroll
    for (int kk = 0; kk < 8; ++kk) { asm volatile("" : "+v"(qh[kk])); asm volatile("" : "+v"(qm[kk])); asm volatile("" : "+v"(ql[kk])); }

    f32x16 s;                                      // S^T of the current tile, then its softmax numerators
    auto s_phase = [&](const unsigned char* cK, int kt) __attribute__((always_inline)) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const int off = k_base ^ (32 * kk);
                const bf16x8 kh = *reinterpret_cast<const bf16x8*>(cK + off);
                const bf16x8 km = *reinterpret_cast<const bf16x8*>(cK + X_KB + off);
                const bf16x8 kl = *reinterpret_cast<const bf16x8*>(cK + 2 * X_KB + off);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kl, qh[kk], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, ql[kk], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qm[kk], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(km, qh[kk], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qm[kk], s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kh, qh[kk], s, 0, 0, 0);
            }
            if ((kt + 1) * TK > NK) {                           // rows past the last key: out of the softmax
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int krow = (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (kt * TK + krow >= NK) s[r] = -1e30f;
                }
            }
            float mx = s[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
            {
                const unsigned mb = __builtin_bit_cast(unsigned, mx);
                const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
            }
            const float m_new = fmaxf(m_run, mx * qscale);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            float psum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[r], qscale, -m_new));
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
    };
    auto pv_phase = [&](const unsigned char* cV) __attribute__((always_inline)) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 ph, pm, pl;
                const f32x4 p0 = {s[kk * 8 + 0], s[kk * 8 + 1], s[kk * 8 + 2], s[kk * 8 + 3]};
                const f32x4 p1 = {s[kk * 8 + 4], s[kk * 8 + 5], s[kk * 8 + 6], s[kk * 8 + 7]};
                x_split8(p0, p1, ph, pm, pl);
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const unsigned char* vp = cV + (v_base ^ (64 * dt)) + kk * (16 * 256);
                    bf16x8 v[3];
#pragma unroll
                    for (int p3 = 0; p3 < 3; ++p3) {
                        const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((x_lds_s16x4*)(vp + p3 * X_KB));
                        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((x_lds_s16x4*)(vp + p3 * X_KB + 8 * 256));
                        const s16x8 av = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
                        v[p3] = __builtin_bit_cast(bf16x8, av);
                    }
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[2], ph, acc[dt], 0, 0, 0);
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[0], pl, acc[dt], 0, 0, 0);
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[1], pm, acc[dt], 0, 0, 0);
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[1], ph, acc[dt], 0, 0, 0);
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[0], pm, acc[dt], 0, 0, 0);
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[0], ph, acc[dt], 0, 0, 0);
                }
            }
    };
    if constexpr (KS == 1) {
        issue_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int cur = 0;
        for (int kt = 0; kt < ntiles; ++kt) {
            const unsigned char* cK = smem + cur * X_STAGE;
            if (kt + 1 < ntiles) issue_tile(kt + 1, cur ^ 1);       // lands under this tile's products
            if (wave_active) {
                s_phase(cK, kt);
                pv_phase(cK + 3 * X_KB);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of tile kt + 1 have landed
            __syncthreads();
            cur ^= 1;
        }
    } else {
        // ring of this group: K stage 0 | K stage 1 | V   (3 planes of 8 KB each)
        const unsigned char* const ring = smem + kg * RING;
        issue_half(kg, 0u, false);
        issue_half(kg, 6u * X_KB, true);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int cur = 0;
        // round 6: the two key groups (waves w and w + NW share a SIMD) ONE PHASE APART: while group 0 is in its S phase (QK^T MFMAs, then the
        // softmax's VALU) group 1 is in its PV phase (the split's VALU, then the PV MFMAs) and vice versa -- complementary halves per SIMD
        // (conv_bf16x.hip PP, MI355X_MICROARCH.md "Two waves per SIMD").  The rings are per group, the barriers per workgroup: group 1
        // passes one extra barrier in front of its loop, group 0 one behind its loop; nothing else moves.  Same bits; 165.8 -> 159.4 us
        // per block at the headline shape (tools/probe/att_ab.py).
        if (kg == 1) __syncthreads();
        for (int itr = 0; itr < nIter; ++itr) {
            const int kt = kg + itr * KS;                             // may be == ntiles for the last group: an all-masked tile
            const bool more = itr + 1 < nIter;
            if (more) issue_half(kt + KS, (unsigned)((cur ^ 1) * 3 * X_KB), false);       // K of the next tile: lands under this tile
            if (wave_active) s_phase(ring + cur * 3 * X_KB, kt);
            // this tile's V planes were issued BEFORE the K planes above (loads return in order): all but those may be pending
            if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                                          // ... everybody's pieces of V
            if (wave_active) pv_phase(ring + 6 * X_KB);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of the next K
            __syncthreads();                                          // V and K[cur] are free, K[cur ^ 1] is complete
            if (more) issue_half(kt + KS, 6u * X_KB, true);           // V of the next tile: lands under its S products and softmax
            cur ^= 1;
        }
        if (kg == 0) __syncthreads();
        // merge the key groups: O = sum_g 2^(m_g - m) O_g, l likewise (attention.hip); group 1 parks its state in its own ring
        float* scr = reinterpret_cast<float*>(smem + RING);
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
        {
            const float m_o = scr[tid], l_o = scr[NT + tid];
            const float m_new = fmaxf(m_run, m_o);
            const float a0 = __builtin_amdgcn_exp2f(m_run - m_new), a1 = __builtin_amdgcn_exp2f(m_o - m_new);
            l_run = l_run * a0 + l_o * a1;
            m_run = m_new;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[dt][r] = acc[dt][r] * a0 + scr[(2 + dt * 16 + r) * NT + tid] * a1;
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
            for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    const f32x4 v = {acc[dt][rq * 4 + 0] * inv, acc[dt][rq * 4 + 1] * inv, acc[dt][rq * 4 + 2] * inv, acc[dt][rq * 4 + 3] * inv};
                    *reinterpret_cast<f32x4*>(op + dt * 32 + 8 * rq + 4 * h) = v;
                }
        }
    }
#endif
}

}  // namespace

/* k / v columns of `rows` consecutive fp32 qkv rows (token rows followed by the pooled rows) -> three bf16 planes whose sum
 * is the fp32 value bit for bit: planes[p][row][1024], p = hi, mid, lo */
extern "C" int e2fgvi_split3_kv(const float* qkv_rows, void* planes, int64_t rows, void* stream) {
    E2_REQUIRE(qkv_rows && planes && rows > 0, E2FGVI_EINVAL, "split3_kv: null pointer / no rows");
    E2_REQUIRE(((uintptr_t)qkv_rows & 15) == 0 && ((uintptr_t)planes & 15) == 0, E2FGVI_EINVAL, "split3_kv: buffers must be 16-byte aligned");
    const long long items = rows * (CP / 8);
    hipLaunchKernelGGL(split3_kv_kernel, dim3((unsigned)cdiv64(items, 256)), dim3(256), 0, (hipStream_t)stream, qkv_rows,
                       (unsigned short*)planes, (long long)rows);
    E2_LAUNCH_CHECK("split3_kv");
    return 0;
}

/* e2fgvi_focal_attention with both products on the bf16 matrix pipe (six exact bf16 terms per fp32 product).  qkv: the fp32
 * token rows [B*T*fh*fw][1536] (read for Q only); planes: e2fgvi_split3_kv of those rows FOLLOWED by the B*T*nWin pooled rows;
 * out fp32 [rows][512].  waves: 0 (auto), 2, 4 or 8 waves of 32 queries per workgroup; 14 = four waves x two key groups. */
extern "C" int e2fgvi_focal_attention_x3(const float* qkv, const void* planes, const int32_t* key_tab, int32_t tab_ld,
                                         const int32_t* nkeys, float* out, int32_t B, int32_t T, int32_t fh, int32_t fw,
                                         int32_t waves, void* stream) {
    E2_REQUIRE(qkv && planes && key_tab && nkeys && out, E2FGVI_EINVAL, "focal_attention_x3: null pointer");
    E2_REQUIRE(B > 0 && T > 0 && fh > 0 && fw > 0 && fh % WS0 == 0 && fw % WS1 == 0, E2FGVI_EINVAL,
               "focal_attention_x3: token grid %dx%d must be a positive multiple of (5,9)", fh, fw);
    E2_REQUIRE(tab_ld >= SLOTS, E2FGVI_EINVAL, "focal_attention_x3: tab_ld < 210");
    E2_REQUIRE(((uintptr_t)qkv & 15) == 0 && ((uintptr_t)planes & 15) == 0 && ((uintptr_t)out & 15) == 0, E2FGVI_EINVAL,
               "focal_attention_x3: buffers must be 16-byte aligned");
    const int nWin = (fh / WS0) * (fw / WS1);
    const long long rows = (long long)B * T * fh * fw + (long long)B * T * nWin;
    const long long plane_bytes = rows * CP * 2;
    E2_REQUIRE(3 * plane_bytes < 0xFFFFF000LL, E2FGVI_EUNSUP, "focal_attention_x3: the three k / v planes span >= 4 GiB (split the batch)");
    // waves: 0 (auto), 2 / 4 / 8 = waves of 32 queries per workgroup, one key group; 14 = four waves x TWO key groups
    int ks = 1;
    if (waves == 14) { ks = 2; waves = 4; }
    const int ntiles = cdiv(T * SLOTS, TK);
    size_t dyn = (size_t)ntiles * TK * 4;
    const size_t dyn2 = (size_t)cdiv(ntiles, 2) * 2 * TK * 4;
    const bool fits2 = dyn2 + 2 * 9 * X_KB + 1024 + 256 <= 160 * 1024;
    E2_REQUIRE(dyn + 2 * X_STAGE + 1024 + 256 <= 160 * 1024, E2FGVI_EUNSUP, "focal_attention_x3: window of %d frames does not fit the LDS key table", T);
    if (waves <= 0) {
        // enough workgroups for the chip first, then as many queries per staged K / V tile as possible; when there are not even
        // two 32-query blocks per SIMD (the 432x240 clip: 900 for 1024), two key groups per workgroup
        const long long wg4 = (long long)cdiv(T * WTOK, 128) * nWin * NH * B;
        const long long qblocks = (long long)cdiv(T * WTOK, 32) * nWin * NH * B;
        waves = wg4 >= 512 ? 8 : 4;
        if (wg4 < 192) waves = 2;
        if (waves == 4 && qblocks < 2048 && fits2) ks = 2;
    }
    E2_REQUIRE(waves == 2 || waves == 4 || waves == 8, E2FGVI_EINVAL, "focal_attention_x3: waves must be 0, 2, 4, 8 or 14");
    E2_REQUIRE(ks == 1 || fits2, E2FGVI_EUNSUP, "focal_attention_x3: two key groups do not fit the LDS for a window of %d frames", T);
    if (ks == 2) dyn = dyn2;
    const long long nblk = (long long)cdiv(T * WTOK, 32 * waves) * nWin * NH * B;
    E2_REQUIRE(nblk < 2147483647LL, E2FGVI_EUNSUP, "focal_attention_x3: more than 2^31 workgroups");
    dim3 grid((unsigned)nblk), block(64 * waves * ks);
#define E2_ATT_X3(NW_, KS_)                                                                                                       \
    do {                                                                                                                          \
        hipError_t ea = hipFuncSetAttribute((const void*)focal_attn_x3_kernel<NW_, KS_>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                            (int)dyn);                                                                            \
        E2_REQUIRE(ea == hipSuccess, (int)ea, "focal_attention_x3: cannot reserve %zu bytes of dynamic LDS", dyn);                \
        hipLaunchKernelGGL((focal_attn_x3_kernel<NW_, KS_>), grid, block, dyn, (hipStream_t)stream, qkv, key_tab, tab_ld, nkeys, out, \
                           B, T, fh, fw, (const char*)planes, (unsigned)(3 * plane_bytes), (unsigned)plane_bytes,                 \
                           (unsigned)((long long)B * T * fh * fw));                                                               \
    } while (0)
    if (ks == 2) E2_ATT_X3(4, 2);
    else if (waves == 2) E2_ATT_X3(2, 1);
    else if (waves == 4) E2_ATT_X3(4, 1);
    else E2_ATT_X3(8, 1);
#undef E2_ATT_X3
    E2_LAUNCH_CHECK("focal_attention_x3");
    return 0;
}
