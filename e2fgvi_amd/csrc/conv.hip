// Implicit-GEMM NHWC convolution / linear layer on fp32 MFMA for gfx950.
//
//   out[m][n] = sum_k A[m][k] * W[k][n],   m = output pixel (img,oy,ox), n = output channel,
//   k = (tap ky,kx ; channel of the virtual concat of up to 4 sources), per group.
//
// Tiling: a workgroup owns a BM x BN output tile; waves form a WGM x WGN grid, each accumulating
// TM x TN 32x32 MFMA tiles in registers.  K advances in BK-channel chunks inside one tap of one
// source: the A slab of a chunk is BM pixels x BK contiguous floats (NHWC: one 64/128-byte run per
// pixel, coalesced), gathered with the im2col address computed on the fly -- nothing is
// materialised.  Global -> registers -> LDS with a 2-deep LDS ring: the loads of chunk k+1 are in
// flight while chunk k is multiplied; one barrier per chunk.
// Weights are pre-packed as [group][k/4][Npad][4] so a lane's 16-byte load/LDS read carries 4
// consecutive k of one output channel (see mma_ktile in common.h).
// Epilogue fuses bias, residual add, activation and the channel-slice / NCHW store.
//
// Reference operator calls replaced: see include/e2fgvi_hip.h (e2fgvi_conv2d_nhwc).
#include "common.h"

namespace {

struct ConvParams {
    const float* src[E2FGVI_MAX_SRC];
    int ld[E2FGVI_MAX_SRC];
    int coff[E2FGVI_MAX_SRC];
    int cpg[E2FGVI_MAX_SRC];
    int nsrc;
    int N, H, W, Ho, Wo, KH, KW, stride, pad;
    int Cout, Cout_g, Npad;
    int M;
    int tilesM, tilesN;
    int chunks_per_tap;
    unsigned src_bytes[E2FGVI_MAX_SRC];   // extent of each source (buffer bounds; < 4 GiB)
    unsigned wgroup_bytes;
    long long wgroup_stride;   // floats per group in the packed weight
    const float* w;
    const float* bias;
    const float* res;
    int res_ld, res_coff;
    float* dst;
    int dst_ld, dst_coff, dst_nchw;
    int act;
    float slope;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// E2FGVI_ACT_DCNPOST: SecondOrderDeformableAlignment's post-processing of the conv_offset output (reference
// model/modules/feat_prop.py:38-53) fused into the producing conv: channels [0, 2C/3) are (dy,dx)-interleaved offsets
// -> max_residue * tanh(v) + flow (dy takes the flow's y component; first half of the offsets uses flow_1, second half
// flow_2); channels [2C/3, C) are masks -> sigmoid(v).  fl = this pixel's (u1, v1, u2, v2).
__device__ __forceinline__ float dcn_post(float v, int co, int C, const float* fl, float max_residue) {
    const int noff = (C / 3) * 2;
    if (co >= noff) return e2_fast_sigmoid(v);
    const int which = (co * 2 >= noff) ? 2 : 0;
    const float f = fl[which + ((co & 1) ? 0 : 1)];       // even channel = dy <- v (index 1), odd = dx <- u (index 0)
    return max_residue * e2_fast_tanh(v) + f;
}

// raw buffer resource: out-of-range offsets (>= bytes) return 0 -- the hardware does the zero padding of the
// convolution halo, of the padded channels and of the K tail, with no branch and no select on the loaded data
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}

template <int BM, int BN, int BK, int WGM, int WGN, int D, int KS>
__global__ __launch_bounds__(64 * WGM * WGN * KS) void conv_igemm_kernel(const ConvParams p) {
    // KS > 1: intra-workgroup split-K.  KS groups of WGM x WGN waves work on the SAME output tile, group kg taking
    // K-chunks kg, kg+KS, ... with its own LDS ring; partial sums are merged through LDS at the end.  It is how a
    // small-M problem (one output tile per CU, e.g. the 6480-pixel propagation convs) gets more than one wave per
    // SIMD, so that one group's loads / LDS traffic / barrier waits hide under the other's MFMAs.
    constexpr int NG = 64 * WGM * WGN;    // threads per K-group
    constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
    constexpr int LDA = BK + 4;
    constexpr int CH = BK / 4;            // float4 chunks per A row
    constexpr int RP = NG / CH;           // A rows covered per pass
    constexpr int A_IT = (BM + RP - 1) / RP;
    constexpr int B_F4 = BK * BN / 4;
    constexpr int B_IT = (B_F4 + NG - 1) / NG;
    constexpr int STAGE = BM * LDA + BK * BN;      // floats of one LDS stage (A slab + B slab)
    constexpr unsigned OOB = 0xFFFFFFFFu;
    static_assert(TM >= 1 && TN >= 1, "tile");
    static_assert(NG % CH == 0, "threads per row");
    static_assert(KS * 2 * STAGE >= (KS - 1) * NG * TM * TN * 16, "reduction scratch must fit in LDS");

    __shared__ __attribute__((aligned(16))) float smem[KS * 2 * STAGE];

    // K-group: wave-uniform; readfirstlane tells the compiler so, which keeps the whole K-loop state (tap, source,
    // channel) and the per-source parameters it indexes in SGPRs
    const int kg = (KS == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x / NG));
    const int tid = threadIdx.x - kg * NG;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int g = blockIdx.y;
    float* sbase = smem + kg * (2 * STAGE);

    const int logical = xcd_remap(blockIdx.x, p.tilesM * p.tilesN);
    const int tile_m = logical / p.tilesN, tile_n = logical - tile_m * p.tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread A-row bookkeeping (fixed for the whole K loop): pixel index of tap (0,0) and its coordinates;
    //      rows outside the problem get coordinates that fail every bounds test
    const int c4 = tid % CH;
    int a_pix[A_IT], a_by[A_IT], a_bx[A_IT];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int ia = 0; ia < A_IT; ++ia) {
        const int row = tid / CH + ia * RP;
        const int m = m0 + row;
        const bool ok = (row < BM) && (m < p.M);
        const int mm = ok ? m : 0;
        const int img = mm / HoWo;
        const int rem = mm - img * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_by[ia] = ok ? oy * p.stride - p.pad : -(1 << 28);
        a_bx[ia] = ox * p.stride - p.pad;
        a_pix[ia] = (img * p.H + oy * p.stride - p.pad) * p.W + ox * p.stride - p.pad;
    }
    // weights: this thread's fixed part of the byte offset, OOB for columns past Npad / rows past the slab
    unsigned b_off[B_IT];
#pragma unroll
    for (int ib = 0; ib < B_IT; ++ib) {
        const int f = tid + ib * NG;
        const int kq = f / BN, n = f - kq * BN;
        const bool ok = (B_F4 % NG == 0 || f < B_F4) && (n0 + n) < p.Npad;
        b_off[ib] = ok ? (unsigned)((kq * p.Npad + n0 + n) * 16) : OOB;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w + (long long)g * p.wgroup_stride, p.wgroup_bytes);
    const unsigned b_step = (unsigned)(BK / 4) * (unsigned)p.Npad * 16u;     // bytes per K-chunk

    // ---- K-loop state (wave-uniform): the next chunk this group will load
    int ky = 0, kx = 0, s = 0, c0 = 0;
    const int KT = p.KH * p.KW * p.chunks_per_tap;

    // D register stages: this group's next D chunks are in flight while the current one is multiplied
    f32x4 ra[D][A_IT], rb[D][B_IT];

    // parameters of the source being walked, carried in scalar registers and re-read from the kernel arguments only when
    // the walk moves to another source (indexing p.src[s] ... per chunk costs two dependent scalar-memory round trips)
    const float* cur_src = p.src[0];
    unsigned cur_bytes = p.src_bytes[0];
    unsigned cur_ld4 = (unsigned)p.ld[0] * 4u;
    unsigned cur_chan = (unsigned)(p.coff[0] + g * p.cpg[0]) * 4u;
    int cur_cpg = p.cpg[0];

    auto load_tile = [&](int kt, f32x4 (&qa)[A_IT], f32x4 (&qb)[B_IT]) {
        const bool tile_ok = kt < KT;                                   // past the end: everything reads as zero
        const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(cur_src, cur_bytes);
        const unsigned ld4 = cur_ld4;
        const unsigned chan = cur_chan + (unsigned)(c0 + c4 * 4) * 4u;
        const bool cok = tile_ok && (c0 + c4 * 4) < cur_cpg;
        const int tap = ky * p.W + kx;
#pragma unroll
        for (int ia = 0; ia < A_IT; ++ia) {
            const bool ok = cok && (unsigned)(a_by[ia] + ky) < (unsigned)p.H && (unsigned)(a_bx[ia] + kx) < (unsigned)p.W;
            const unsigned off = (unsigned)(a_pix[ia] + tap) * ld4 + chan;
            qa[ia] = buf_load4(arsrc, ok ? off : OOB);
        }
        const unsigned wk = tile_ok ? (unsigned)kt * b_step : OOB;
#pragma unroll
        for (int ib = 0; ib < B_IT; ++ib) {
            const unsigned off = b_off[ib] + wk;                         // OOB + x may wrap: keep it saturated
            qb[ib] = buf_load4(wrsrc, (b_off[ib] == OOB || !tile_ok) ? OOB : off);
        }
    };
    auto store_tile = [&](int buf, const f32x4 (&qa)[A_IT], const f32x4 (&qb)[B_IT]) {
        float* sA = sbase + buf * STAGE;
        float* sB = sA + BM * LDA;
#pragma unroll
        for (int ia = 0; ia < A_IT; ++ia) {
            const int row = tid / CH + ia * RP;
            if (BM % RP == 0 || row < BM) *reinterpret_cast<f32x4*>(sA + row * LDA + c4 * 4) = qa[ia];
        }
#pragma unroll
        for (int ib = 0; ib < B_IT; ++ib) {
            const int f = tid + ib * NG;
            if (B_F4 % NG == 0 || f < B_F4) *reinterpret_cast<f32x4*>(sB + f * 4) = qb[ib];
        }
    };
    auto advance1 = [&]() {
        c0 += BK;
        if (c0 >= cur_cpg) {
            c0 = 0;
            ++s;
            if (s == p.nsrc) {
                s = 0;
                ++kx;
                if (kx == p.KW) { kx = 0; ++ky; }
            }
            if (p.nsrc > 1) {
                cur_src = p.src[s]; cur_bytes = p.src_bytes[s]; cur_ld4 = (unsigned)p.ld[s] * 4u;
                cur_chan = (unsigned)(p.coff[s] + g * p.cpg[s]) * 4u; cur_cpg = p.cpg[s];
            }
        }
    };
    auto advance = [&]() {
#pragma unroll
        for (int q = 0; q < KS; ++q) advance1();
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // this group's chunks are kt = kg + it*KS.  The loop below has NO guards: chunks past the end load zeros (buffer
    // bounds) and multiply zeros, so the iteration count is simply rounded up to a multiple of D.
    for (int q = 0; q < kg; ++q) advance1();
    const int nIter = ((KT + KS - 1) / KS + D - 1) / D * D;

    // prologue: chunk it=0 -> LDS[0]; chunks it=1..D-1 -> register stages 1..D-1 (stage index = it % D)
    load_tile(kg, ra[0], rb[0]);
    advance();
    store_tile(0, ra[0], rb[0]);
#pragma unroll
    for (int j = 1; j < D; ++j) {
        load_tile(kg + j * KS, ra[j], rb[j]);
        advance();
    }
    __syncthreads();

    int cur = 0;
    for (int it0 = 0; it0 < nIter; it0 += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int kt = kg + (it0 + j) * KS;
            load_tile(kt + D * KS, ra[j], rb[j]);      // stage j held chunk kt (already in LDS): refill it
            advance();
            mma_ktile<TM, TN, BK, LDA, BN>(sbase + cur * STAGE, sbase + cur * STAGE + BM * LDA, acc, wm * TM * 32,
                                           wn * TN * 32, lane);
            store_tile(cur ^ 1, ra[(j + 1) % D], rb[(j + 1) % D]);
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- merge the K-groups' partial sums through LDS (the rings are free after the last barrier)
    if (KS > 1) {
        constexpr int PART = NG * TM * TN * 16;
        if (kg > 0) {
            float* sq = smem + (kg - 1) * PART;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) sq[((tm * TN + tn) * 16 + r) * NG + tid] = acc[tm][tn][r];
        }
        __syncthreads();
        if (kg > 0) return;
#pragma unroll
        for (int q = 1; q < KS; ++q) {
            const float* sq = smem + (q - 1) * PART;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[tm][tn][r] += sq[((tm * TN + tn) * 16 + r) * NG + tid];
        }
    }

    // ---- epilogue
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + (wn * TN + tn) * 32 + j;
        if (n >= p.Cout_g) continue;
        const int co = g * p.Cout_g + n;
        const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int m = m0 + (wm * TM + tm) * 32 + row;
                if (m >= p.M) continue;
                float v = acc[tm][tn][r] + bv;
                if (p.act == E2FGVI_ACT_DCNPOST) v = dcn_post(v, co, p.Cout, p.res + (long long)m * 4, p.slope);
                else {
                    if (p.res) v += p.res[(long long)m * p.res_ld + p.res_coff + co];
                    v = apply_act(v, p.act, p.slope);
                }
                if (p.dst_nchw) {
                    const int img = m / HoWo, rem = m - img * HoWo;
                    p.dst[((long long)img * p.Cout + co) * HoWo + rem] = v;
                } else {
                    p.dst[(long long)m * p.dst_ld + p.dst_coff + co] = v;
                }
            }
        }
    }
}

// ---- halo-staged variant ---------------------------------------------------------------------------
// Stride-1 KxK convolutions on large images re-read every input pixel K*K times through the implicit-GEMM path; the
// re-reads miss the 4 MiB L2 (a layer's weights alone fill it) and show up as fabric traffic.  Here a workgroup owns
// a TH x 16 pixel tile of ONE image: per channel block the (TH+K-1) x (16+K-1) input patch is staged once in LDS and
// all K*K taps take their A operand from it at a shifted pixel offset (one scalar add per tap); only the weight slab
// of each tap streams through the 2-deep LDS ring.  Global A traffic drops ~6x (3x3) / ~20x (7x7), the per-chunk
// address arithmetic disappears, the MFMA work and the packed weights are unchanged (chunk (tap, block) of the
// [tap][channel] packing is simply visited in (block, tap) order).
// TG = taps per weight stage: 1 (one barrier per tap) or KK_ (a whole kernel row per barrier -- narrow layers do only
// CB/2 MFMAs per wave and tap, too little to amortise a barrier).
template <int KK_, int TH, int BN, int CB, int WGM, int WGN, int TG>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_halo_kernel(const ConvParams p) {
    static_assert(TG == 1 || TG == KK_, "TG");
    constexpr int TW = 16;
    constexpr int PH = TH + KK_ - 1, PW = TW + KK_ - 1, NPIX = PH * PW;
    constexpr int NT = 64 * WGM * WGN;
    constexpr int TM = (TH / 2) / WGM, TN = BN / (32 * WGN);
    constexpr int LDP = CB + 4;
    constexpr int CH = CB / 4;
    constexpr int P_F4 = NPIX * CH;
    constexpr int P_IT = (P_F4 + NT - 1) / NT;
    constexpr int B1_F4 = CB * BN / 4;          // float4s of one tap's weight slab
    constexpr int B_F4 = TG * B1_F4;
    constexpr int B_IT = (B_F4 + NT - 1) / NT;
    constexpr unsigned OOB = 0xFFFFFFFFu;
    static_assert(TM >= 1 && TN >= 1 && (TH / 2) % WGM == 0, "tile");

    __shared__ __attribute__((aligned(16))) float smem[2 * NPIX * LDP + 2 * TG * CB * BN];
    float* sP0 = smem;
    float* sB0 = smem + 2 * NPIX * LDP;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int g = blockIdx.y;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int nsp = p.N * tiles_y * tiles_x;
    const int logical = xcd_remap(blockIdx.x, nsp * p.tilesN);
    const int tile_n = logical % p.tilesN;
    int sp = logical / p.tilesN;
    const int txi = sp % tiles_x;
    sp /= tiles_x;
    const int tyi = sp % tiles_y;
    const int img = sp / tiles_y;
    const int y0 = tyi * TH, x0 = txi * TW, n0 = tile_n * BN;

    // patch staging: this thread's float4s (pixel of the patch, channel quad): input pixel index or "outside"
    int pp_pix[P_IT];         // (img*H + iy)*W + ix, or -1
    int pp_lds[P_IT];         // float offset inside the patch buffer
    int pp_c4[P_IT];
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
        const int f = tid + it * NT;
        const int pp = f / CH, c4 = f - pp * CH;
        const int py = pp / PW, px = pp - py * PW;
        const int iy = y0 - p.pad + py, ix = x0 - p.pad + px;
        const bool ok = (P_F4 % NT == 0 || f < P_F4) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        pp_pix[it] = ok ? (img * p.H + iy) * p.W + ix : -1;
        pp_lds[it] = pp * LDP + c4 * 4;
        pp_c4[it] = c4;
    }
    unsigned b_off[B_IT];     // byte offset inside one tap's slab; b_tap = which tap of the stage
    int b_tap[B_IT];
#pragma unroll
    for (int ib = 0; ib < B_IT; ++ib) {
        const int f = tid + ib * NT;
        const int tg = f / B1_F4, f1 = f - tg * B1_F4;
        const int kq = f1 / BN, n = f1 - kq * BN;
        const bool ok = (B_F4 % NT == 0 || f < B_F4) && (n0 + n) < p.Npad;
        b_off[ib] = ok ? (unsigned)((kq * p.Npad + n0 + n) * 16) : OOB;
        b_tap[ib] = tg;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w + (long long)g * p.wgroup_stride, p.wgroup_bytes);
    const unsigned b_step = (unsigned)(CB / 4) * (unsigned)p.Npad * 16u;

    // MFMA A operand: lane (i,h) of sub-tile (wm*TM+tm) is output pixel (2*(wm*TM+tm) + (i>>4), i&15) of the tile
    const int i = lane & 31, h = lane >> 5;
    int a_base[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) a_base[tm] = ((2 * (wm * TM + tm) + (i >> 4)) * PW + (i & 15)) * LDP + h * 4;

    f32x4 rp[P_IT], rb[B_IT];
    const int nblk = p.chunks_per_tap;       // channel blocks over all sources
    int s = 0, c0 = 0;                       // source / first channel of the block being LOADED

    // current source's parameters in scalar registers (see conv_igemm_kernel)
    const float* cur_src = p.src[0];
    unsigned cur_bytes = p.src_bytes[0];
    unsigned cur_ld4 = (unsigned)p.ld[0] * 4u;
    unsigned cur_chan = (unsigned)(p.coff[0] + g * p.cpg[0]) * 4u;
    int cur_cpg = p.cpg[0];
    auto load_patch = [&](bool valid) {
        const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(cur_src, cur_bytes);
        const unsigned ld4 = cur_ld4;
        const unsigned chan = cur_chan + (unsigned)c0 * 4u;
        const int cpg = cur_cpg;
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const bool ok = valid && pp_pix[it] >= 0 && (c0 + pp_c4[it] * 4) < cpg;
            rp[it] = buf_load4(arsrc, ok ? (unsigned)pp_pix[it] * ld4 + chan + (unsigned)pp_c4[it] * 16u : OOB);
        }
    };
    auto store_patch = [&](float* dst) {
#pragma unroll
        for (int it = 0; it < P_IT; ++it)
            if (P_F4 % NT == 0 || (tid + it * NT) < P_F4) *reinterpret_cast<f32x4*>(dst + pp_lds[it]) = rp[it];
    };
    auto advance_blk = [&]() {
        c0 += CB;
        if (c0 >= cur_cpg) {
            c0 = 0; ++s; if (s == p.nsrc) s = 0;
            if (p.nsrc > 1) {
                cur_src = p.src[s]; cur_bytes = p.src_bytes[s]; cur_ld4 = (unsigned)p.ld[s] * 4u;
                cur_chan = (unsigned)(p.coff[s] + g * p.cpg[s]) * 4u; cur_cpg = p.cpg[s];
            }
        }
    };
    // weight stage = taps [tap0, tap0+TG) of channel block `blk`; chunk index in the packing is tap*nblk + blk
    auto load_b = [&](int tap0, int blk, bool valid) {
#pragma unroll
        for (int ib = 0; ib < B_IT; ++ib) {
            const unsigned wk = (unsigned)((tap0 + b_tap[ib]) * nblk + blk) * b_step;
            rb[ib] = buf_load4(wrsrc, (b_off[ib] == OOB || !valid) ? OOB : b_off[ib] + wk);
        }
    };
    auto store_b = [&](float* dst) {
#pragma unroll
        for (int ib = 0; ib < B_IT; ++ib) {
            const int f = tid + ib * NT;
            if (B_F4 % NT == 0 || f < B_F4) *reinterpret_cast<f32x4*>(dst + f * 4) = rb[ib];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    constexpr int NTAP = KK_ * KK_;
    // prologue: patch of block 0 and the weight slab of (block 0, tap 0)
    load_patch(true);
    advance_blk();
    load_b(0, 0, true);
    store_patch(sP0);
    store_b(sB0);
    __syncthreads();

    int pcur = 0, bcur = 0;
    for (int blk = 0; blk < nblk; ++blk) {
        load_patch(blk + 1 < nblk);            // next block's patch is in flight during this block's taps
        advance_blk();
        const float* sP = sP0 + pcur * (NPIX * LDP);
        for (int tap0 = 0; tap0 < NTAP; tap0 += TG) {
            // next weight stage: (blk, tap0+TG) or (blk+1, 0)
            const bool last = tap0 + TG >= NTAP;
            const int ntap0 = last ? 0 : tap0 + TG, nb = last ? blk + 1 : blk;
            load_b(ntap0, nb, nb < nblk);
            const float* sBst = sB0 + bcur * (TG * CB * BN);
#pragma unroll
            for (int tg = 0; tg < TG; ++tg) {
                const int tap = tap0 + tg;
                const int ky = (TG == 1) ? tap / KK_ : tap0 / KK_, kx = (TG == 1) ? tap - ky * KK_ : tg;
                const float* sB = sBst + tg * (CB * BN);
                const int tapoff = (ky * PW + kx) * LDP;
#pragma unroll
                for (int m8 = 0; m8 < CB / 8; ++m8) {
                    f32x4 a[TM], b[TN];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) a[tm] = *reinterpret_cast<const f32x4*>(sP + a_base[tm] + tapoff + m8 * 8);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        b[tn] = *reinterpret_cast<const f32x4*>(sB + ((2 * m8 + h) * BN + (wn * TN + tn) * 32 + i) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn)
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][e], b[tn][e], acc[tm][tn], 0, 0, 0);
                }
            }
            store_b(sB0 + (bcur ^ 1) * (TG * CB * BN));
            if (last) store_patch(sP0 + (pcur ^ 1) * (NPIX * LDP));
            __syncthreads();
            bcur ^= 1;
        }
        pcur ^= 1;
    }

    // ---- epilogue
    const int j = lane & 31;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + (wn * TN + tn) * 32 + j;
        if (n >= p.Cout_g) continue;
        const int co = g * p.Cout_g + n;
        const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;            // pixel of the 2x16 sub-tile
                const int oy = y0 + 2 * (wm * TM + tm) + (row >> 4), ox = x0 + (row & 15);
                if (oy >= p.Ho || ox >= p.Wo) continue;
                const long long m = ((long long)img * p.Ho + oy) * p.Wo + ox;
                float v = acc[tm][tn][r] + bv;
                if (p.act == E2FGVI_ACT_DCNPOST) v = dcn_post(v, co, p.Cout, p.res + m * 4, p.slope);
                else {
                    if (p.res) v += p.res[m * p.res_ld + p.res_coff + co];
                    v = apply_act(v, p.act, p.slope);
                }
                if (p.dst_nchw)
                    p.dst[((long long)img * p.Cout + co) * ((long long)p.Ho * p.Wo) + (long long)oy * p.Wo + ox] = v;
                else
                    p.dst[m * p.dst_ld + p.dst_coff + co] = v;
            }
        }
    }
}

// 16-output-channel variant of the halo kernel on v_mfma_f32_16x16x4_f32 (lane l: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15],
// D[row=(l>>4)*4+reg][col=l&15]): layers with <= 16 output channels (SPyNet 32->16 and 16->2, the 64->3 decoder tail,
// the flow heads) would spend >= half of a 32-wide MFMA on padding.  A wave still owns a 2x16-pixel sub-tile, now as
// two 16-pixel row tiles; each quarter-wave takes 4 consecutive k with one ds_read_b128 (4 MFMAs per read).
template <int KK_, int TH, int CB, int WGM, int TG>
__global__ __launch_bounds__(64 * WGM) void conv_halo16_kernel(const ConvParams p) {
    static_assert(TG == 1 || TG == KK_, "TG");
    static_assert(CB % 16 == 0, "16 k per MFMA step group");
    constexpr int TW = 16, BN = 16;
    constexpr int PH = TH + KK_ - 1, PW = TW + KK_ - 1, NPIX = PH * PW;
    constexpr int NT = 64 * WGM;
    constexpr int TM = (TH / 2) / WGM;
    constexpr int LDP = CB + 4;
    constexpr int CH = CB / 4;
    constexpr int P_F4 = NPIX * CH;
    constexpr int P_IT = (P_F4 + NT - 1) / NT;
    constexpr int B1_F4 = CB * BN / 4;
    constexpr int B_F4 = TG * B1_F4;
    constexpr int B_IT = (B_F4 + NT - 1) / NT;
    constexpr unsigned OOB = 0xFFFFFFFFu;
    static_assert(TM >= 1 && (TH / 2) % WGM == 0, "tile");

    __shared__ __attribute__((aligned(16))) float smem[2 * NPIX * LDP + 2 * TG * CB * BN];
    float* sP0 = smem;
    float* sB0 = smem + 2 * NPIX * LDP;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wm = tid >> 6;
    const int g = blockIdx.y;
    const int tiles_x = (p.Wo + TW - 1) / TW, tiles_y = (p.Ho + TH - 1) / TH;
    const int nsp = p.N * tiles_y * tiles_x;
    int sp = xcd_remap(blockIdx.x, nsp);
    const int txi = sp % tiles_x;
    sp /= tiles_x;
    const int tyi = sp % tiles_y;
    const int img = sp / tiles_y;
    const int y0 = tyi * TH, x0 = txi * TW;

    int pp_pix[P_IT], pp_lds[P_IT], pp_c4[P_IT];
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
        const int f = tid + it * NT;
        const int pp = f / CH, c4 = f - pp * CH;
        const int py = pp / PW, px = pp - py * PW;
        const int iy = y0 - p.pad + py, ix = x0 - p.pad + px;
        const bool ok = (P_F4 % NT == 0 || f < P_F4) && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        pp_pix[it] = ok ? (img * p.H + iy) * p.W + ix : -1;
        pp_lds[it] = pp * LDP + c4 * 4;
        pp_c4[it] = c4;
    }
    unsigned b_off[B_IT];
    int b_tap[B_IT];
#pragma unroll
    for (int ib = 0; ib < B_IT; ++ib) {
        const int f = tid + ib * NT;
        const int tg = f / B1_F4, f1 = f - tg * B1_F4;
        const int kq = f1 / BN, n = f1 - kq * BN;
        const bool ok = (B_F4 % NT == 0 || f < B_F4) && n < p.Npad;
        b_off[ib] = ok ? (unsigned)((kq * p.Npad + n) * 16) : OOB;
        b_tap[ib] = tg;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w + (long long)g * p.wgroup_stride, p.wgroup_bytes);
    const unsigned b_step = (unsigned)(CB / 4) * (unsigned)p.Npad * 16u;

    const int i = lane & 15, kgrp = lane >> 4;
    int a_base[TM][2];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 2; ++r) a_base[tm][r] = ((2 * (wm * TM + tm) + r) * PW + i) * LDP + kgrp * 4;

    f32x4 rp[P_IT], rb[B_IT];
    const int nblk = p.chunks_per_tap;
    int s = 0, c0 = 0;
    // current source's parameters in scalar registers (see conv_igemm_kernel)
    const float* cur_src = p.src[0];
    unsigned cur_bytes = p.src_bytes[0];
    unsigned cur_ld4 = (unsigned)p.ld[0] * 4u;
    unsigned cur_chan = (unsigned)(p.coff[0] + g * p.cpg[0]) * 4u;
    int cur_cpg = p.cpg[0];
    auto load_patch = [&](bool valid) {
        const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(cur_src, cur_bytes);
        const unsigned ld4 = cur_ld4;
        const unsigned chan = cur_chan + (unsigned)c0 * 4u;
        const int cpg = cur_cpg;
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const bool ok = valid && pp_pix[it] >= 0 && (c0 + pp_c4[it] * 4) < cpg;
            rp[it] = buf_load4(arsrc, ok ? (unsigned)pp_pix[it] * ld4 + chan + (unsigned)pp_c4[it] * 16u : OOB);
        }
    };
    auto store_patch = [&](float* dst) {
#pragma unroll
        for (int it = 0; it < P_IT; ++it)
            if (P_F4 % NT == 0 || (tid + it * NT) < P_F4) *reinterpret_cast<f32x4*>(dst + pp_lds[it]) = rp[it];
    };
    auto advance_blk = [&]() {
        c0 += CB;
        if (c0 >= cur_cpg) {
            c0 = 0; ++s; if (s == p.nsrc) s = 0;
            if (p.nsrc > 1) {
                cur_src = p.src[s]; cur_bytes = p.src_bytes[s]; cur_ld4 = (unsigned)p.ld[s] * 4u;
                cur_chan = (unsigned)(p.coff[s] + g * p.cpg[s]) * 4u; cur_cpg = p.cpg[s];
            }
        }
    };
    auto load_b = [&](int tap0, int blk, bool valid) {
#pragma unroll
        for (int ib = 0; ib < B_IT; ++ib) {
            const unsigned wk = (unsigned)((tap0 + b_tap[ib]) * nblk + blk) * b_step;
            rb[ib] = buf_load4(wrsrc, (b_off[ib] == OOB || !valid) ? OOB : b_off[ib] + wk);
        }
    };
    auto store_b = [&](float* dst) {
#pragma unroll
        for (int ib = 0; ib < B_IT; ++ib) {
            const int f = tid + ib * NT;
            if (B_F4 % NT == 0 || f < B_F4) *reinterpret_cast<f32x4*>(dst + f * 4) = rb[ib];
        }
    };

    f32x4 acc[TM][2];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 2; ++r) acc[tm][r] = f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int NTAP = KK_ * KK_;
    load_patch(true);
    advance_blk();
    load_b(0, 0, true);
    store_patch(sP0);
    store_b(sB0);
    __syncthreads();

    int pcur = 0, bcur = 0;
    for (int blk = 0; blk < nblk; ++blk) {
        load_patch(blk + 1 < nblk);
        advance_blk();
        const float* sP = sP0 + pcur * (NPIX * LDP);
        for (int tap0 = 0; tap0 < NTAP; tap0 += TG) {
            const bool last = tap0 + TG >= NTAP;
            const int ntap0 = last ? 0 : tap0 + TG, nb = last ? blk + 1 : blk;
            load_b(ntap0, nb, nb < nblk);
            const float* sBst = sB0 + bcur * (TG * CB * BN);
#pragma unroll
            for (int tg = 0; tg < TG; ++tg) {
                const int tap = tap0 + tg;
                const int ky = (TG == 1) ? tap / KK_ : tap0 / KK_, kx = (TG == 1) ? tap - ky * KK_ : tg;
                const float* sB = sBst + tg * (CB * BN);
                const int tapoff = (ky * PW + kx) * LDP;
#pragma unroll
                for (int m16 = 0; m16 < CB / 16; ++m16) {
                    const f32x4 b = *reinterpret_cast<const f32x4*>(sB + ((4 * m16 + kgrp) * BN + i) * 4);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int r = 0; r < 2; ++r) {
                            const f32x4 a = *reinterpret_cast<const f32x4*>(sP + a_base[tm][r] + tapoff + m16 * 16);
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                acc[tm][r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[e], b[e], acc[tm][r], 0, 0, 0);
                        }
                }
            }
            store_b(sB0 + (bcur ^ 1) * (TG * CB * BN));
            if (last) store_patch(sP0 + (pcur ^ 1) * (NPIX * LDP));
            __syncthreads();
            bcur ^= 1;
        }
        pcur ^= 1;
    }

    const int n = lane & 15;
    if (n < p.Cout_g) {
        const int co = g * p.Cout_g + n;
        const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int oy = y0 + 2 * (wm * TM + tm) + r, ox = x0 + (lane >> 4) * 4 + e;
                    if (oy >= p.Ho || ox >= p.Wo) continue;
                    const long long m = ((long long)img * p.Ho + oy) * p.Wo + ox;
                    float v = acc[tm][r][e] + bv;
                    if (p.act == E2FGVI_ACT_DCNPOST) v = dcn_post(v, co, p.Cout, p.res + m * 4, p.slope);
                    else {
                        if (p.res) v += p.res[m * p.res_ld + p.res_coff + co];
                        v = apply_act(v, p.act, p.slope);
                    }
                    if (p.dst_nchw)
                        p.dst[((long long)img * p.Cout + co) * ((long long)p.Ho * p.Wo) + (long long)oy * p.Wo + ox] = v;
                    else
                        p.dst[m * p.dst_ld + p.dst_coff + co] = v;
                }
    }
}

template <int KK_, int TH, int CB, int WGM, int TG>
int launch_halo16(ConvParams& p, int groups, hipStream_t st) {
    p.tilesN = 1;
    const long long nb = (long long)p.N * cdiv(p.Ho, TH) * cdiv(p.Wo, 16);
    hipLaunchKernelGGL((conv_halo16_kernel<KK_, TH, CB, WGM, TG>), dim3((unsigned)nb, groups, 1), dim3(64 * WGM), 0, st, p);
    E2_LAUNCH_CHECK("conv_halo16");
    return 0;
}

template <int KK_, int TH, int BN, int CB, int WGM, int WGN, int TG>
int launch_halo(ConvParams& p, int groups, hipStream_t st) {
    p.tilesN = cdiv(p.Cout_g, BN);
    const long long nb = (long long)p.N * cdiv(p.Ho, TH) * cdiv(p.Wo, 16) * p.tilesN;
    dim3 grid((unsigned)nb, groups, 1), block(64 * WGM * WGN, 1, 1);
    hipLaunchKernelGGL((conv_halo_kernel<KK_, TH, BN, CB, WGM, WGN, TG>), grid, block, 0, st, p);
    E2_LAUNCH_CHECK("conv_halo");
    return 0;
}

// halo tile codes = 10000 + id:   id, K, TH, BN, CB, WGM, WGN, taps per weight stage
#define E2_HALO_CONFIGS(X)                                                                                          \
    X(1, 3, 8, 128, 32, 2, 2, 1) X(2, 3, 8, 128, 16, 2, 2, 1) X(3, 3, 4, 128, 32, 1, 4, 1) X(4, 3, 4, 128, 16, 1, 4, 1)  \
    X(5, 3, 8, 64, 32, 2, 2, 1) X(11, 3, 8, 64, 16, 2, 2, 1) X(6, 3, 8, 32, 32, 4, 1, 1) X(12, 3, 8, 32, 16, 4, 1, 1)    \
    X(7, 7, 8, 32, 16, 4, 1, 1) X(8, 7, 8, 32, 32, 4, 1, 1) X(9, 7, 8, 64, 32, 2, 2, 1) X(10, 7, 8, 64, 16, 2, 2, 1)     \
    X(22, 3, 8, 128, 16, 2, 2, 3) X(24, 3, 4, 128, 16, 1, 4, 3) X(31, 3, 8, 64, 16, 2, 2, 3) X(25, 3, 8, 64, 32, 2, 2, 3) \
    X(32, 3, 8, 32, 16, 4, 1, 3) X(26, 3, 8, 32, 32, 4, 1, 3)                                                        \
    X(27, 7, 8, 32, 16, 4, 1, 7) X(30, 7, 8, 64, 16, 2, 2, 7) X(28, 7, 8, 32, 32, 4, 1, 7)                          \
    X(51, 7, 8, 32, 8, 4, 1, 7) X(52, 3, 8, 32, 8, 4, 1, 3) X(53, 3, 8, 64, 8, 2, 2, 3) X(54, 7, 8, 64, 8, 2, 2, 7)

// 16-wide halo tiles: id, K, TH, CB, WGM, taps per weight stage
#define E2_HALO16_CONFIGS(X) \
    X(41, 7, 8, 16, 4, 7) X(42, 3, 8, 16, 4, 3) X(43, 7, 8, 32, 4, 7) X(44, 3, 8, 32, 4, 3) X(45, 3, 8, 16, 4, 1) X(46, 7, 8, 16, 4, 1)

int halo_cb_of(int id) {
    switch (id) {
#define X(id_, k, th, cb, wm, tg) case id_: return cb;
        E2_HALO16_CONFIGS(X)
#undef X
#define X(id_, k, th, bn, cb, wm, wn, tg) case id_: return cb;
        E2_HALO_CONFIGS(X)
#undef X
        default: return 0;
    }
}
int halo_k_of(int id) {
    switch (id) {
#define X(id_, k, th, cb, wm, tg) case id_: return k;
        E2_HALO16_CONFIGS(X)
#undef X
#define X(id_, k, th, bn, cb, wm, wn, tg) case id_: return k;
        E2_HALO_CONFIGS(X)
#undef X
        default: return 0;
    }
}
int dispatch_halo(ConvParams& p, int groups, int id, hipStream_t st) {
    switch (id) {
#define X(id_, k, th, cb, wm, tg) case id_: return launch_halo16<k, th, cb, wm, tg>(p, groups, st);
        E2_HALO16_CONFIGS(X)
#undef X
#define X(id_, k, th, bn, cb, wm, wn, tg) case id_: return launch_halo<k, th, bn, cb, wm, wn, tg>(p, groups, st);
        E2_HALO_CONFIGS(X)
#undef X
        default: break;
    }
    e2fgvi_set_error("conv2d: halo tile id %d is not instantiated", id);
    return E2FGVI_EINVAL;
}

// ---- weight packing ---------------------------------------------------------------------------
struct PackParams {
    int Cout, groups, KH, KW, nsrc, bk;
    int cpg[E2FGVI_MAX_SRC];
    int Cout_g, Npad, Cin_g, chunks_per_tap;
    long long total;          // floats
    long long wgroup_stride;
};

__global__ void pack_conv_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, const PackParams p) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.total) return;
    const int g = (int)(idx / p.wgroup_stride);
    long long rem = idx - (long long)g * p.wgroup_stride;
    const int e = (int)(rem & 3);
    rem >>= 2;
    const int n = (int)(rem % p.Npad);
    const int kq = (int)(rem / p.Npad);
    const int k = kq * 4 + e;
    const int kt = k / p.bk, kk = k - kt * p.bk;
    const int tap = kt / p.chunks_per_tap;
    int chunk = kt - tap * p.chunks_per_tap;
    int s = 0, prefix = 0;
    while (true) {
        const int nc = (p.cpg[s] + p.bk - 1) / p.bk;
        if (chunk < nc) break;
        chunk -= nc;
        prefix += p.cpg[s];
        ++s;
    }
    const int c = chunk * p.bk + kk;
    float v = 0.f;
    if (c < p.cpg[s] && n < p.Cout_g) {
        const int cin = prefix + c;
        v = w[((long long)(g * p.Cout_g + n) * p.Cin_g + cin) * (p.KH * p.KW) + tap];
    }
    wp[idx] = v;
}

bool geometry(int Cout, int groups, int KH, int KW, int nsrc, const int32_t* cpg, int bk, PackParams* q) {
    if (Cout <= 0 || groups <= 0 || Cout % groups || KH <= 0 || KW <= 0 || nsrc < 1 || nsrc > E2FGVI_MAX_SRC) return false;
    if (bk != 8 && bk != 16 && bk != 32) return false;
    q->Cout = Cout; q->groups = groups; q->KH = KH; q->KW = KW; q->nsrc = nsrc; q->bk = bk;
    q->Cout_g = Cout / groups;
    q->Npad = round_up(q->Cout_g, 32);
    q->Cin_g = 0;
    q->chunks_per_tap = 0;
    for (int s = 0; s < E2FGVI_MAX_SRC; ++s) q->cpg[s] = 0;
    for (int s = 0; s < nsrc; ++s) {
        if (cpg[s] <= 0 || cpg[s] % 4) return false;
        q->cpg[s] = cpg[s];
        q->Cin_g += cpg[s];
        q->chunks_per_tap += cdiv(cpg[s], bk);
    }
    q->wgroup_stride = (long long)KH * KW * q->chunks_per_tap * bk * q->Npad;
    q->total = q->wgroup_stride * groups;
    return true;
}

template <int BM, int BN, int BK, int WGM, int WGN, int D, int KS>
int launch_conv(ConvParams& p, int groups, hipStream_t st) {
    p.tilesM = cdiv(p.M, BM);
    p.tilesN = cdiv(p.Cout_g, BN);
    dim3 grid(p.tilesM * p.tilesN, groups, 1), block(64 * WGM * WGN * KS, 1, 1);
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WGM, WGN, D, KS>), grid, block, 0, st, p);
    E2_LAUNCH_CHECK("conv_igemm");
    return 0;
}

// tile code = shape + 10 * (kernel K-chunk: 1=16, 2=32, 3=64, 4=8) + 100 * register stages + 1000 * (K-groups - 1)
// shapes: 1 128x128  2 128x64  3 64x64  4 128x32  5 64x32  6 64x128  7 32x128  8 256x128  9 128x256 (8 waves)
#define E2_CONV_CONFIGS(X)                                                                                \
    X(121, 128, 128, 32, 2, 2, 1, 1) X(221, 128, 128, 32, 2, 2, 2, 1) X(211, 128, 128, 16, 2, 2, 2, 1)     \
    X(111, 128, 128, 16, 2, 2, 1, 1) X(1111, 128, 128, 16, 2, 2, 1, 2) X(1121, 128, 128, 32, 2, 2, 1, 2)   \
    X(222, 128, 64, 32, 2, 2, 2, 1) X(212, 128, 64, 16, 2, 2, 2, 1) X(122, 128, 64, 32, 2, 2, 1, 1)        \
    X(1122, 128, 64, 32, 2, 2, 1, 2) X(1222, 128, 64, 32, 2, 2, 2, 2)                                      \
    X(123, 64, 64, 32, 2, 2, 1, 1) X(223, 64, 64, 32, 2, 2, 2, 1) X(233, 64, 64, 64, 2, 2, 2, 1)           \
    X(213, 64, 64, 16, 2, 2, 2, 1) X(1223, 64, 64, 32, 2, 2, 2, 2) X(1123, 64, 64, 32, 2, 2, 1, 2)         \
    X(2123, 64, 64, 32, 2, 2, 1, 3) X(2223, 64, 64, 32, 2, 2, 2, 3) X(3123, 64, 64, 32, 2, 2, 1, 4)        \
    X(1233, 64, 64, 64, 2, 2, 2, 2) X(1133, 64, 64, 64, 2, 2, 1, 2)                                        \
    X(1213, 64, 64, 16, 2, 2, 2, 2) X(2213, 64, 64, 16, 2, 2, 2, 3) X(3213, 64, 64, 16, 2, 2, 2, 4)        \
    X(224, 128, 32, 32, 4, 1, 2, 1) X(214, 128, 32, 16, 4, 1, 2, 1) X(1224, 128, 32, 32, 4, 1, 2, 2)       \
    X(225, 64, 32, 32, 2, 1, 2, 1) X(215, 64, 32, 16, 2, 1, 2, 1) X(1225, 64, 32, 32, 2, 1, 2, 2)          \
    X(3225, 64, 32, 32, 2, 1, 2, 4) X(1215, 64, 32, 16, 2, 1, 2, 2) X(3215, 64, 32, 16, 2, 1, 2, 4)        \
    X(226, 64, 128, 32, 2, 2, 2, 1) X(216, 64, 128, 16, 2, 2, 2, 1) X(1226, 64, 128, 32, 2, 2, 2, 2)       \
    X(1126, 64, 128, 32, 2, 2, 1, 2) X(126, 64, 128, 32, 2, 2, 1, 1)                                       \
    X(227, 32, 128, 32, 1, 4, 2, 1) X(217, 32, 128, 16, 1, 4, 2, 1) X(1227, 32, 128, 32, 1, 4, 2, 2)       \
    X(2227, 32, 128, 32, 1, 4, 2, 3)                                                                       \
    X(218, 256, 128, 16, 4, 2, 2, 1) X(228, 256, 128, 32, 4, 2, 2, 1) X(118, 256, 128, 16, 4, 2, 1, 1)     \
    X(219, 128, 256, 16, 2, 4, 2, 1) X(119, 128, 256, 16, 2, 4, 1, 1)                                      \
    X(241, 128, 128, 8, 2, 2, 2, 1) X(242, 128, 64, 8, 2, 2, 2, 1) X(243, 64, 64, 8, 2, 2, 2, 1)           \
    X(244, 128, 32, 8, 4, 1, 2, 1) X(245, 64, 32, 8, 2, 1, 2, 1) X(246, 64, 128, 8, 2, 2, 2, 1)            \
    X(3245, 64, 32, 8, 2, 1, 2, 4)

int dispatch_tile(ConvParams& p, int groups, int code, hipStream_t st) {
    switch (code) {
#define X(id, bm, bn, bk, wm, wn, d, ks) \
    case id: return launch_conv<bm, bn, bk, wm, wn, d, ks>(p, groups, st);
        E2_CONV_CONFIGS(X)
#undef X
        default: break;
    }
    e2fgvi_set_error("conv2d: tile code %d is not instantiated", code);
    return E2FGVI_EINVAL;
}

int kernel_bk_of(int code) { const int k = (code / 10) % 10; return k == 1 ? 16 : k == 2 ? 32 : k == 3 ? 64 : k == 4 ? 8 : 0; }

int auto_shape(const ConvParams& p, int groups) {
    auto blocks = [&](int bm, int bn) { return (long long)cdiv(p.M, bm) * cdiv(p.Cout_g, bn) * groups; };
    const long long want = 2 * 256;    // >= 2 workgroups per CU before growing the tile
    if (p.Cout_g <= 32) return blocks(128, 32) >= want ? 4 : 5;
    if (p.Cout_g <= 64) return blocks(128, 64) >= want ? 2 : 3;
    if (blocks(128, 128) >= want) {
        // 8-wave 256x128 tiles halve the weight-slab traffic per flop: +2...27 % measured when they still give
        // every CU ~2 workgroups
        if (p.Cout_g >= 128 && blocks(256, 128) >= 400) return 8;
        return 1;
    }
    if (blocks(64, 128) >= want) return 6;
    return 3;
}

}  // namespace

// This file is compiled twice (e2fgvi_amd/build.py): once normally -> e2fgvi_conv2d_nhwc, and once with
// -DE2_NOPK_VARIANT and the packed-fp32 target feature removed -> e2fgvi_conv2d_nhwc_nopk, the entry point of the layers
// that run on a side stream next to bf16 MFMA tiles (SPyNet): packed-fp32 VALU instructions consuming freshly loaded
// registers are corrupted in lanes 48-63 beside those tiles (DESIGN.md "Stream overlap", tools/probe/overlap_probe.hip).
#ifndef E2_NOPK_VARIANT
extern "C" int64_t e2fgvi_packed_conv_weight_size(int32_t Cout, int32_t groups, int32_t KH, int32_t KW,
                                                  int32_t nsrc, const int32_t* src_cpg, int32_t bk) {
    PackParams q;
    if (!src_cpg || !geometry(Cout, groups, KH, KW, nsrc, src_cpg, bk, &q)) {
        e2fgvi_set_error("packed_conv_weight_size: bad geometry");
        return E2FGVI_EINVAL;
    }
    return q.total;
}

extern "C" int e2fgvi_pack_conv_weight(const float* w, float* wpacked, int32_t Cout, int32_t groups, int32_t KH,
                                       int32_t KW, int32_t nsrc, const int32_t* src_cpg, int32_t bk, void* stream) {
    PackParams q;
    E2_REQUIRE(w && wpacked && src_cpg, E2FGVI_EINVAL, "pack_conv_weight: null pointer");
    E2_REQUIRE(geometry(Cout, groups, KH, KW, nsrc, src_cpg, bk, &q), E2FGVI_EINVAL, "pack_conv_weight: bad geometry");
    const int nt = 256;
    hipLaunchKernelGGL(pack_conv_weight_kernel, dim3((unsigned)cdiv64(q.total, nt)), dim3(nt), 0, (hipStream_t)stream,
                       w, wpacked, q);
    E2_LAUNCH_CHECK("pack_conv_weight");
    return 0;
}

#endif

#ifdef E2_NOPK_VARIANT
extern "C" int e2fgvi_conv2d_nhwc_nopk(const e2fgvi_conv_desc* d, void* stream) {
#else
extern "C" int e2fgvi_conv2d_nhwc(const e2fgvi_conv_desc* d, void* stream) {
#endif
    E2_REQUIRE(d, E2FGVI_EINVAL, "conv2d: null descriptor");
    PackParams q;
    E2_REQUIRE(geometry(d->Cout, d->groups, d->KH, d->KW, d->nsrc, d->src_cpg, d->bk, &q), E2FGVI_EINVAL,
               "conv2d: bad geometry (Cout %d groups %d k %dx%d nsrc %d bk %d)", d->Cout, d->groups, d->KH, d->KW,
               d->nsrc, d->bk);
    E2_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->stride > 0 && d->pad >= 0, E2FGVI_EINVAL,
               "conv2d: bad sizes");
    E2_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1,
               E2FGVI_EINVAL, "conv2d: Ho/Wo inconsistent with H/W/k/stride/pad");
    E2_REQUIRE((long long)d->N * d->Ho * d->Wo < 2147483647LL, E2FGVI_EUNSUP, "conv2d: more than 2^31 output pixels");
    E2_REQUIRE(d->wpacked && d->dst, E2FGVI_EINVAL, "conv2d: null weight/dst");
    ConvParams p;
    for (int s = 0; s < E2FGVI_MAX_SRC; ++s) {
        p.src[s] = nullptr; p.ld[s] = 0; p.coff[s] = 0; p.cpg[s] = 0;
    }
    for (int s = 0; s < d->nsrc; ++s) {
        E2_REQUIRE(d->src[s], E2FGVI_EINVAL, "conv2d: null source %d", s);
        E2_REQUIRE(d->src_ld[s] % 4 == 0 && d->src_coff[s] % 4 == 0 && ((uintptr_t)d->src[s] & 15) == 0, E2FGVI_EINVAL,
                   "conv2d: source %d not 16-byte addressable (ld %d coff %d)", s, d->src_ld[s], d->src_coff[s]);
        E2_REQUIRE(d->src_coff[s] + d->groups * d->src_cpg[s] <= d->src_ld[s], E2FGVI_EINVAL,
                   "conv2d: source %d channel range exceeds its pixel stride", s);
        p.src[s] = d->src[s]; p.ld[s] = d->src_ld[s]; p.coff[s] = d->src_coff[s]; p.cpg[s] = d->src_cpg[s];
        const long long bytes = (long long)d->N * d->H * d->W * d->src_ld[s] * 4;
        E2_REQUIRE(bytes < 4294967295LL, E2FGVI_EUNSUP,
                   "conv2d: source %d spans %lld bytes; buffer addressing needs < 4 GiB (split the batch)", s, bytes);
        p.src_bytes[s] = (unsigned)bytes;
    }
    for (int s = d->nsrc; s < E2FGVI_MAX_SRC; ++s) p.src_bytes[s] = 0;
    E2_REQUIRE(q.wgroup_stride * 4 < 4294967295LL, E2FGVI_EUNSUP, "conv2d: packed weight group >= 4 GiB");
    p.wgroup_bytes = (unsigned)(q.wgroup_stride * 4);
    E2_REQUIRE(((uintptr_t)d->wpacked & 15) == 0, E2FGVI_EINVAL, "conv2d: packed weight not 16-byte aligned");
    p.nsrc = d->nsrc;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
    p.Cout = d->Cout; p.Cout_g = q.Cout_g; p.Npad = q.Npad;
    p.M = d->N * d->Ho * d->Wo;
    p.chunks_per_tap = q.chunks_per_tap;
    p.wgroup_stride = q.wgroup_stride;
    p.w = d->wpacked; p.bias = d->bias; p.res = d->residual; p.res_ld = d->res_ld; p.res_coff = d->res_coff;
    p.dst = d->dst; p.dst_ld = d->dst_ld; p.dst_coff = d->dst_coff; p.dst_nchw = d->dst_nchw;
    p.act = d->act; p.slope = d->slope;
    if (d->act == E2FGVI_ACT_DCNPOST)
        E2_REQUIRE(d->residual && d->res_ld == 4 && d->Cout % 3 == 0 && d->groups == 1 && !d->dst_nchw, E2FGVI_EINVAL,
                   "conv2d: ACT_DCNPOST needs residual = flows [P,4], Cout multiple of 3, groups 1, NHWC output");
    if (!d->dst_nchw)
        E2_REQUIRE(d->dst_coff >= 0 && d->dst_coff + d->Cout <= d->dst_ld, E2FGVI_EINVAL, "conv2d: dst slice exceeds dst_ld");
    int code = d->tile;
    if (code == 0 && d->stride == 1 && d->KH == d->KW && (d->KH == 3 || d->KH == 7) && q.Cout_g <= 64) {
        // narrow stride-1 layers on large images (decoder tail, SPyNet top levels): the halo-staged kernel wins
        // 5...12 % (measured, tools/conv_bench.py) and cuts the im2col re-read traffic; wide layers stay on the
        // implicit-GEMM path, which is MFMA-bound and faster there
        const long long tiles = (long long)d->N * cdiv(d->Ho, 8) * cdiv(d->Wo, 16);
        bool c16 = true;     // a 16-channel block must be the pack granule or divide every source without padding
        for (int s = 0; s < d->nsrc; ++s) c16 = c16 && (d->bk == 16 || d->src_cpg[s] % 32 == 0);
        bool c8 = true;      // ... and an 8-channel block likewise
        for (int s = 0; s < d->nsrc; ++s) c8 = c8 && d->src_cpg[s] % 8 == 0 && (d->bk == 8 || d->src_cpg[s] % d->bk == 0);
        if (d->KH == 7) {
            // SPyNet's 7x7 stacks (8 -> 32 -> 64 -> 32 -> 16 -> 2 on 18 frame pairs, 64x128 ... 2x4 images), measured per level
            // with tools/spynet_bench.py (profiles/r02_spynet_bench.txt): the 16-wide halo tile for <= 16 output channels at
            // EVERY level (73 vs 30 us on 32x64), the 8-channel-block halo tile 51 for wider outputs down to 32x64 images
            // (the former rule left those to the implicit GEMM: 98 -> 78, 133 -> 102 us; on 64x128 it beats the former
            // choices 27 / 10 by 6-22 %), four K-groups of the implicit GEMM for the 64-channel layer on the tiny levels.
            // SPyNet runs beside the encoder but is not hidden by it: its microseconds count.
            int cin = 0;
            for (int s = 0; s < d->nsrc; ++s) cin += d->src_cpg[s];
            if (q.Cout_g <= 16 && c16) code = 10041;
            else if (c8 && (tiles >= 128 || cin <= 8 || (cin <= 32 && tiles >= 64))) code = 10051;
            else if (d->bk == 32 && cin % 32 == 0 && tiles < 128) code = 3225;
        } else if (tiles >= 512 && d->Wo >= 32 && c16) {
            const bool narrow = q.Cout_g <= 32;
            code = 10000 + (narrow ? 12 : 11);
            if (q.Cout_g <= 16) code = 10042;                                       // 16-wide MFMA tiles
        } else if (tiles >= 512 && d->Wo >= 32 && d->bk == 8) {                     // 4/8-channel inputs
            const bool narrow = q.Cout_g <= 32;
            code = 10000 + (narrow ? 52 : 53);
        }
    }
    if (code >= 10000) {   // halo-staged kernel
        const int id = code - 10000;
        const int cb = halo_cb_of(id), k = halo_k_of(id);
        E2_REQUIRE(cb != 0, E2FGVI_EINVAL, "conv2d: bad halo tile id %d", id);
        E2_REQUIRE(d->stride == 1 && d->KH == k && d->KW == k, E2FGVI_EINVAL,
                   "conv2d: halo tile %d needs a stride-1 %dx%d convolution", id, k, k);
        E2_REQUIRE(id < 40 || id > 50 || q.Cout_g <= 16, E2FGVI_EINVAL, "conv2d: halo tile %d is for <= 16 output channels per group", id);
        if (cb != d->bk)
            for (int s = 0; s < d->nsrc; ++s)
                E2_REQUIRE(d->src_cpg[s] % cb == 0 && d->src_cpg[s] % d->bk == 0, E2FGVI_EINVAL,
                           "conv2d: halo tile %d (channel block %d) incompatible with source %d of %d channels", id, cb, s,
                           d->src_cpg[s]);
        p.chunks_per_tap = 0;
        for (int s = 0; s < d->nsrc; ++s) p.chunks_per_tap += cdiv(d->src_cpg[s], cb);
        return dispatch_halo(p, d->groups, id, (hipStream_t)stream);
    }
    if (code < 10) {     // shape only (or auto): pick K-chunk, register stages and K-groups from the measurements
        const int shape = code ? code : auto_shape(p, d->groups);
        bool all16 = true;
        for (int s = 0; s < d->nsrc; ++s) all16 = all16 && (d->src_cpg[s] % 32 == 0);
        int kb = d->bk;
        // >= 128-wide tiles: a 16-deep K-chunk halves the LDS ring, so 4 workgroups fit per CU (+4...7 % measured)
        if ((shape == 1 || shape == 2 || shape == 6 || shape == 8) && d->bk == 32 && all16) kb = 16;
        int ks = 1;
        if (shape == 3) {
            // one output tile per CU or less (e.g. the 6480-pixel propagation convs): 3 K-groups per workgroup
            // give every SIMD three waves to overlap loads / LDS / barriers with MFMAs (-20...30 % measured)
            const long long blocks = (long long)cdiv(p.M, 64) * cdiv(p.Cout_g, 64) * d->groups;
            int kt = 0;
            for (int s = 0; s < d->nsrc; ++s) kt += cdiv(d->src_cpg[s], kb);
            kt *= d->KH * d->KW;
            if (blocks < 384 && kt >= 24) ks = 3;
        }
        if (shape == 5) {
            // narrow, tiny-M layers (the low SPyNet pyramid levels: 144 ... 9216 pixels): a handful of workgroups
            // walking up to 98 K-chunks each -- pure latency; 4 K-groups cut the serial chain by 4
            const long long blocks = (long long)cdiv(p.M, 64) * cdiv(p.Cout_g, 32) * d->groups;
            int kt = 0;
            for (int s = 0; s < d->nsrc; ++s) kt += cdiv(d->src_cpg[s], kb);
            kt *= d->KH * d->KW;
            if (blocks < 384 && kt >= 24) ks = 4;
        }
        int shp = shape;
        if (kb == 8 && shp == 8) shp = 1;                 // no 8-wave tile for 8-channel chunks
        if (kb == 8 && shp == 3) ks = 1;                  // (the 64x64 K-group merge scratch needs a deeper chunk)
        code = (ks - 1) * 1000 + 200 + (kb == 8 ? 40 : kb == 16 ? 10 : 20) + shp;
    }
    const int kbk = kernel_bk_of(code);
    E2_REQUIRE(kbk != 0, E2FGVI_EINVAL, "conv2d: bad tile code %d", code);
    if (kbk != d->bk) {  // a different K-chunk is only legal when no source needs padding under either granule
        for (int s = 0; s < d->nsrc; ++s)
            E2_REQUIRE(d->src_cpg[s] % kbk == 0 && d->src_cpg[s] % d->bk == 0, E2FGVI_EINVAL,
                       "conv2d: tile code %d (K-chunk %d) incompatible with source %d of %d channels", code, kbk, s,
                       d->src_cpg[s]);
    }
    // chunks_per_tap follows the kernel's K-chunk
    p.chunks_per_tap = 0;
    for (int s = 0; s < d->nsrc; ++s) p.chunks_per_tap += cdiv(d->src_cpg[s], kbk);
    return dispatch_tile(p, d->groups, code, (hipStream_t)stream);
}
