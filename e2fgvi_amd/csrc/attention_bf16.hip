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


// ------------------------------------------------------------------------------------------------------------------------
// Round 3: the same operator with the staging work taken out of the waves' instruction streams.
//
// Round 2's kernel above spends most of its issue slots around the MFMAs, not in them (profiles/r02_layer_table_hq720_bf16.md:
// 590 TF/s): per 32-key tile every thread computes gather addresses (table lookup, pooled / token row selection, an integer
// multiply), stages K and V through registers, and writes V transposed with eight 2-byte ds_writes; its eight waves run in
// lockstep behind one barrier per tile, so the softmax VALU of one wave never overlaps the MFMAs of its SIMD partner.  Here:
//   * the byte offset of EVERY key row of the window (T x nv of them) is computed once per workgroup into an LDS table; a
//     tile's gather is then one ds_read_b32 per 4-row piece;
//   * K and V tiles go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB = 4 key rows per wave instruction):
//     no staging registers, no ds_write pass.  The DMA writes LDS linearly in lane order, so the conflict-free images are
//     produced on the SOURCE side: lane (row, slot) fetches the 16-byte chunk that belongs in its slot
//       K: slot = chunk ^ (row & 15)                    (b128 operand reads down a column of keys, as before)
//       V: slot = ((chunk >> 2) ^ (row & 3)) * 4 + (chunk & 3)   (the four key rows of a transposing read hit four 64-byte
//                                                         bank ranges)
//   * V stays ROW-major in LDS; the PV product's A operand (V^T) is formed by ds_read_b64_tr_b16, gfx950's transposing LDS
//     read: inside a 16-lane group, lane L receives element (L & 3) of the 8-byte words of lanes 4 j + (L >> 2), j = 0..3
//     -- i.e. 4 consecutive keys of ONE d column (tools/probe/tr_read_probe.hip prints the map).  The product's k order is
//     still the order in which the S^T accumulators hold the keys, so P needs no cross-lane traffic;
//   * small workgroups (NW waves x 32 QB queries) of which several are resident per CU: the waves of different workgroups
//     are not tied to each other's barriers, so one's softmax overlaps another's MFMAs;
//   * softmax per element: max, fma (scale and shift folded), exp2, add; the out-of-range mask only in the last tile.
#ifndef E2_ATT_SETPRIO
#define E2_ATT_SETPRIO 0      // experiment switch (tools): s_setprio(1) around the MFMA clusters of the tile loop
#endif
constexpr int V2_KB = TK * HD * 2;                // 8 KB: K tile, [32 keys][16 slots of 16 bytes]
constexpr int V2_VB = TK * HD * 2;                // 8 KB: V tile, row-major, slots permuted per row
typedef __attribute__((address_space(3))) void v2_lds_void;
typedef int v2_i32x4 __attribute__((ext_vector_type(4)));

// One LDS-DMA piece: 64 lanes x 16 bytes from the buffer `rsrc` at the lanes' byte offsets (out-of-range offsets land as
// zeros) to LDS bytes [lds_dst, lds_dst + 1024), lds_dst wave-uniform.  Issued through inline asm on purpose: for the
// builtin (__builtin_amdgcn_raw_ptr_buffer_load_lds) hipcc places an `s_waitcnt vmcnt(0)` in front of the next LDS read that
// may alias the destination -- here the first operand read of the CURRENT tile, a few instructions after the NEXT tile's
// pieces were issued, which exposes the whole L2 latency once per tile.  hipcc does not count an asm load: the kernel waits
// for its pieces itself (vmcnt(0) + barrier at the end of the tile).  M0 (the LDS base of the instruction) is written in the
// same statement that reads it and restored afterwards; `s_nop 4`: descriptor SGPRs fresh from v_readfirstlane
// (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void v2_dma16(v2_i32x4 rsrc, unsigned lds_dst, unsigned voff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ v2_i32x4 v2_rsrc_words(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    v2_i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xFFFFu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 v2_lds_s16x4;

template <int NW, int QB>
__global__ __launch_bounds__(64 * NW, (QB == 1 ? 3 : 2)) void focal_attn_bf16_v2_kernel(const __bf16* __restrict__ qkv, const int* __restrict__ key_tab,
                                                                     int tab_ld, const int* __restrict__ nkeys,
                                                                     __bf16* __restrict__ out, int B, int T, int fh, int fw,
                                                                     const char* lo_base, unsigned lo_bytes, unsigned q_rel,
                                                                     unsigned p_rel, int xcd) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NT = 64 * NW;
    constexpr int PIECES = 8 / NW;                 // 1-KiB DMA pieces of a K (and of a V) tile per wave
    // table entry of a key row past the end: still out of the buffer's range after the in-row byte offset (< 2 KiB) is added,
    // so the DMA lands zeros without a select per piece (the launcher keeps the buffer below 0xFFFFF000 bytes)
    constexpr unsigned OOB = 0xFFFFF000u;
    static_assert(NW == 2 || NW == 4 || NW == 8, "waves per workgroup");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * V2_KB + 2 * V2_VB];
    __shared__ int stab[256];
    extern __shared__ __attribute__((aligned(16))) unsigned ktab[];       // byte offset of every key row, OOB past the end

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int nWw = fw / WS1, nWh = fh / WS0, nWin = nWh * nWw;
    // 1-D grid, query chunk fastest.  Block b runs on XCD b % 8, so the query chunks of one (window, head) -- which read the
    // same K / V rows -- land on DIFFERENT XCDs and each fetches the rows into its own L2 (PMC FETCH_SIZE of this kernel: 8.7 GB
    // per 720p forward against 3.6 for round 2's 256-query workgroups).  Making them neighbours on one XCD (xcd_remap, `xcd` =
    // 1, E2FGVI_ATT_XCD=1) removes the re-fetch but measured SLOWER: 329 -> 366 us per block at 720p T=10, 2634 -> 2737 at
    // 1080p T=20 (profiles/r03_attention_variants.txt) -- four workgroups streaming the same rows at the same time queue on
    // the same L2 channels, and the re-fetches are Infinity-Cache hits anyway (the block's qkv tensor is 199 MB).  Default off.
    const int nqc = (T * WTOK + 32 * QB * NW - 1) / (32 * QB * NW);
    const int logical = xcd ? xcd_remap(blockIdx.x, gridDim.x) : (int)blockIdx.x;
    const int qchunk = logical % nqc;
    const int wh = (logical / nqc) % (nWin * NH);
    const int b = logical / (nqc * nWin * NH);
    const int win = wh / NH, head = wh - win * NH;
    const int wy = win / nWw, wx = win - wy * nWw;
    const int NQ = T * WTOK;
    const int ntok = fh * fw;
    const v2_i32x4 rsrc = v2_rsrc_words(lo_base, lo_bytes);

    const int nv = nkeys[win];
    const int NK = T * nv;
    const int ntiles = (NK + TK - 1) / TK;
    const int* tab = key_tab + (long long)win * tab_ld;
    for (int e = tid; e < nv && e < 256; e += NT) stab[e] = tab[e];
    __syncthreads();
    {   // the key-row table: entry k = (frame t = k / nv, slot s = k % nv), walked without divisions
        int t = 0, sl = tid;
        while (sl >= nv) { sl -= nv; ++t; }
        const unsigned head_off = (unsigned)((512 + head * HD) * 2);
        for (int k = tid; k < ntiles * TK; k += NT) {
            unsigned e = OOB;
            if (k < NK) {
                const int ref = stab[sl];
                const bool pooled = ref < 0;
                const unsigned rowi = pooled ? (unsigned)((b * T + t) * nWin + (-(ref + 1))) : (unsigned)((b * T + t) * ntok + ref);
                e = rowi * (unsigned)(CQ * 2) + head_off + (pooled ? p_rel : q_rel);
            }
            ktab[k] = e;
            sl += NT;
            while (sl >= nv) { sl -= nv; ++t; }
        }
    }
    __syncthreads();                               // the table is read by every wave's DMA address arithmetic

    // ---- this wave's QB x 32 queries: 128 d as 8 operand octets per k-step
    const int q0 = (qchunk * NW + wave) * (32 * QB);
    const bool wave_active = q0 < NQ;
    auto query_row = [&](int j, bool& ok) -> long long {      // token row of this lane's query of block j (recomputed in the
        const int qi = q0 + 32 * j + i;                        // epilogue rather than kept in registers through the tile loop)
        ok = qi < NQ;
        const int qq = ok ? qi : 0;
        const int t = qq / WTOK, pp = qq - t * WTOK;
        const int py = pp / WS1, px = pp - py * WS1;
        return (long long)(b * T + t) * ntok + (wy * WS0 + py) * fw + (wx * WS1 + px);
    };
    bf16x8 q[QB][8];
#pragma unroll
    for (int j = 0; j < QB; ++j) {
        bool ok;
        const long long row = query_row(j, ok);
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *reinterpret_cast<const u32x4*>(qkv + row * CQ + head * HD + kk * 16 + h * 8);
            q[j][kk] = __builtin_bit_cast(bf16x8, v);
        }
    }
    const float qscale = 0.08838834764831845f * LOG2E;       // 128^-0.5 * log2(e)

    f32x16 acc[QB][4];
#pragma unroll
    for (int j = 0; j < QB; ++j)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[j][dt][r] = 0.f;
    float m_run[QB], l_run[QB];
#pragma unroll
    for (int j = 0; j < QB; ++j) { m_run[j] = -1e30f; l_run[j] = 0.f; }

    // ---- DMA bookkeeping: piece pc = wave + NW * jp covers key rows 4 pc .. 4 pc + 3 of a tile; lane -> (row, slot)
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
    // LDS byte address of the stage area (the DMA destination is an address, not a pointer)
    const unsigned smem_lds = (unsigned)(unsigned long long)(v2_lds_void*)smem;
    auto issue_tile = [&](int kt, int stage) {
        const unsigned sk = smem_lds + (unsigned)(stage * V2_KB + wave * 1024);
        const unsigned sv = smem_lds + (unsigned)(2 * V2_KB + stage * V2_VB + wave * 1024);
        unsigned e[PIECES];
#pragma unroll
        for (int jp = 0; jp < PIECES; ++jp) e[jp] = ktab[kt * TK + d_row[jp]];
#pragma unroll
        for (int jp = 0; jp < PIECES; ++jp) {
            v2_dma16(rsrc, __builtin_amdgcn_readfirstlane(sk + jp * NW * 1024), e[jp] + d_kb[jp]);
            v2_dma16(rsrc, __builtin_amdgcn_readfirstlane(sv + jp * NW * 1024), e[jp] + d_vb[jp]);
        }
    };

    // ---- operand read addresses inside a stage
    // K (S^T = K . Q^T, A operand): lane (i = key, h) reads chunk 2 kk + h of key i at slot (2 kk + h) ^ (i & 15)
    // V (O^T += V^T . P, A operand): MFMA kk multiplies keys 16 kk + 4 h + {0..3} (e = 0..3) and + 8 + {0..3} (e = 4..7) of
    // column d = 32 dt + i.  One transposing read per key quad: lane l (lam = l & 15, group g = l >> 4 = 2 h + (i >> 4))
    // supplies the address of V[key quad base + (lam >> 2)][32 dt + 16 (i >> 4) + 4 (lam & 3) .. + 3]
    // Both swizzles are XORs of disjoint bit fields, so ONE base register per operand serves every step:
    //   K slot (2 kk + h) ^ (i & 15) = (2 kk) ^ (h ^ (i & 15))           -> byte address = k_base ^ (32 kk)
    //   V slot ((dt ^ kap) << 2) | chunk_lo                              -> byte address = v_base ^ (64 dt)
    const int lam = lane & 15, kap = lam >> 2;
    const int key_l = 4 * h + kap;                                           // + 16 kk + 8 half: multiples of 4, (key & 3) = kap
    const int chunk_lo = 2 * ((i >> 4) & 1) + ((lam & 3) >> 1);              // 16-byte chunk inside the 64-byte block of dt
    const int v_base = key_l * 256 + ((kap << 2) | chunk_lo) * 16 + (lam & 1) * 8;
    const int k_base = i * 256 + ((h ^ (i & 15)) << 4);

    // the Q rows are ordinary (compiler-counted) loads: make hipcc wait for them HERE -- otherwise its "may still be
    // outstanding" state flows into the tile loop and it waits with vmcnt(0) in front of the loop's first MFMA, i.e. for
    // the next tile's DMA pieces issued a few instructions earlier
#pragma unroll
    for (int j = 0; j < QB; ++j)
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) asm volatile("" : "+v"(q[j][kk]));

    issue_tile(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < ntiles; ++kt) {
        const unsigned char* cK = smem + cur * V2_KB;
        const unsigned char* cV = smem + 2 * V2_KB + cur * V2_VB;
        if (kt + 1 < ntiles) issue_tile(kt + 1, cur ^ 1);       // lands under this tile's products

        if (wave_active) {
            f32x16 s[QB];
#pragma unroll
            for (int j = 0; j < QB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[j][r] = 0.f;
            if (E2_ATT_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                const bf16x8 a = *reinterpret_cast<const bf16x8*>(cK + (k_base ^ (32 * kk)));
#pragma unroll
                for (int j = 0; j < QB; ++j) s[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, q[j][kk], s[j], 0, 0, 0);
            }
            if (E2_ATT_SETPRIO) __builtin_amdgcn_s_setprio(0);
            if (kt == ntiles - 1) {                             // rows past the last key: out of the softmax
#pragma unroll
                for (int j = 0; j < QB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int krow = (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (kt * TK + krow >= NK) s[j][r] = -1e30f;
                    }
            }
            bf16x8 pk[QB][2];
#pragma unroll
            for (int j = 0; j < QB; ++j) {
                float mx = s[j][0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[j][r]);
                {   // the other 16 keys of this query's row sit in lane ^ 32: v_permlane32_swap hands each half the other's value
                    // (no LDS round trip, no lgkmcnt wait that would also drain the V reads in flight)
                    const unsigned mb = __builtin_bit_cast(unsigned, mx);
                    const auto sw = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);
                    mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
                }
                const float m_new = fmaxf(m_run[j], mx * qscale);
                const float alpha = __builtin_amdgcn_exp2f(m_run[j] - m_new);
                float psum = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    s[j][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][r], qscale, -m_new));
                    psum += s[j][r];
                }
                l_run[j] = l_run[j] * alpha + psum;
                m_run[j] = m_new;
                if (__any(alpha != 1.0f)) {
#pragma unroll
                    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[j][dt][r] *= alpha;
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int e = 0; e < 8; ++e) pk[j][kk][e] = (__bf16)s[j][kk * 8 + e];
            }
            if (E2_ATT_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const unsigned char* vp = cV + (v_base ^ (64 * dt)) + kk * (16 * 256);
                    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v2_lds_s16x4*)(vp));
                    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v2_lds_s16x4*)(vp + 8 * 256));
                    typedef short s16x8 __attribute__((ext_vector_type(8)));
                    const s16x8 av = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    const bf16x8 a = __builtin_bit_cast(bf16x8, av);
#pragma unroll
                    for (int j = 0; j < QB; ++j) acc[j][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pk[j][kk], acc[j][dt], 0, 0, 0);
                }
            if (E2_ATT_SETPRIO) __builtin_amdgcn_s_setprio(0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's pieces of tile kt + 1 have landed
        __syncthreads();
        cur ^= 1;
    }

    if (wave_active) {
#pragma unroll
        for (int j = 0; j < QB; ++j) {
            float l = l_run[j] + __shfl_xor(l_run[j], 32);
            const float nmask = (float)(T * (SLOTS - nv));
            l += nmask * __builtin_amdgcn_exp2f(-100.f * LOG2E - m_run[j]);
            const float inv = 1.f / l;
            bool ok;
            const long long row = query_row(j, ok);
            if (ok) {
                __bf16* op = out + row * (NH * HD) + head * HD;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt)
#pragma unroll
                    for (int rq = 0; rq < 4; ++rq) {
                        typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                        bf16x4 v = {(__bf16)(acc[j][dt][rq * 4 + 0] * inv), (__bf16)(acc[j][dt][rq * 4 + 1] * inv),
                                    (__bf16)(acc[j][dt][rq * 4 + 2] * inv), (__bf16)(acc[j][dt][rq * 4 + 3] * inv)};
                        *reinterpret_cast<bf16x4*>(op + dt * 32 + 8 * rq + 4 * h) = v;
                    }
            }
        }
    }
#endif
}

}  // namespace

static int g_att_variant = 0;        // 0: automatic; set through the ABI (tools / tests), never from the environment (round 6)

extern "C" int e2fgvi_focal_attention_bf16_variant(int variant) {
    const int prev = g_att_variant;
    g_att_variant = variant;
    return prev;
}

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
    E2_REQUIRE(hi_end < 0xFFFFF000LL, E2FGVI_EUNSUP,
               "focal_attention_bf16: qkv and kv_pool must lie within one 4 GiB window (allocate them back to back / split the batch)");
    // Variant (e2fgvi_focal_attention_bf16_variant): 10 * QB + NW selects the round-3 kernel with NW waves of QB x 32 queries per
    // workgroup (12, 14, 18, 22, 24); 1 = round 2's kernel; 0 = automatic.
    constexpr int xcd_env = 0;
    int variant = g_att_variant;
    const size_t dyn = (size_t)cdiv(T * SLOTS, TK) * TK * 4;              // the key-row table of the round-3 kernel
    const bool v2_fits = dyn + 2 * V2_KB + 2 * V2_VB + 1024 + 256 <= 160 * 1024;
    if (variant == 0) {
        // measured (profiles/r03_attention_variants.txt): four waves of 32 queries (three workgroups per CU) win on short windows
        // and small grids; two query blocks per wave (each K / V operand read from LDS feeds two MFMAs, and a (window, head)'s
        // K / V rows are fetched by half as many workgroups) win once the grid still fills the chip twice: 1080p T=20 2634 vs
        // 2777 us; 720p T=10 a tie in time at half the K / V fetch traffic
        const long long wgs24 = (long long)cdiv(T * WTOK, 256) * nWin * NH * B;
        variant = !v2_fits ? 1 : (T * WTOK >= 400 && wgs24 >= 1024) ? 24 : 14;
    }
    if (variant != 1 && !v2_fits) variant = 1;                            // very long windows (T > 150): round 2's kernel
    if (variant != 1) {
        const int nw = variant % 10, qb = variant / 10;
        E2_REQUIRE((nw == 2 || nw == 4 || nw == 8) && (qb == 1 || qb == 2), E2FGVI_EINVAL, "focal_attention_bf16: unknown variant %d", variant);
        const long long nblk = (long long)cdiv(T * WTOK, 32 * qb * nw) * nWin * NH * B;
        E2_REQUIRE(nblk < 2147483647LL, E2FGVI_EUNSUP, "focal_attention_bf16: more than 2^31 workgroups");
        dim3 grid((unsigned)nblk), block(64 * nw);
#define E2_ATT_V2(NW_, QB_)                                                                                                       \
        do {                                                                                                                      \
            if (dyn + 2 * V2_KB + 2 * V2_VB + 1024 > 64 * 1024) {                                                                 \
                hipError_t ea = hipFuncSetAttribute((const void*)focal_attn_bf16_v2_kernel<NW_, QB_>,                             \
                                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);                         \
                E2_REQUIRE(ea == hipSuccess, (int)ea, "focal_attention_bf16: cannot reserve %zu bytes of dynamic LDS", dyn);      \
            }                                                                                                                     \
            hipLaunchKernelGGL((focal_attn_bf16_v2_kernel<NW_, QB_>), grid, block, dyn, (hipStream_t)stream, (const __bf16*)qkv, \
                               key_tab, tab_ld, nkeys, (__bf16*)out, B, T, fh, fw, lo, (unsigned)hi_end, (unsigned)(cq - lo),      \
                               (unsigned)(cp - lo), xcd_env);                                                                     \
        } while (0)
        if (qb == 1 && nw == 2) E2_ATT_V2(2, 1);
        else if (qb == 1 && nw == 4) E2_ATT_V2(4, 1);
        else if (qb == 1 && nw == 8) E2_ATT_V2(8, 1);
        else if (qb == 2 && nw == 2) E2_ATT_V2(2, 2);
        else if (qb == 2 && nw == 4) E2_ATT_V2(4, 2);
        else E2_ATT_V2(8, 2);
#undef E2_ATT_V2
        E2_LAUNCH_CHECK("focal_attention_bf16 (v2)");
        return 0;
    }
    // eight query waves per workgroup when a window has enough query tiles (720p T=10: 15): every staged K / V tile then
    // serves 256 queries instead of 128, half the staging work per MFMA (-0.2 ms per 720p forward)
    const int nw = qtiles >= 12 ? 8 : 4;
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
