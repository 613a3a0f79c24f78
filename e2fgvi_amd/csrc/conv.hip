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

template <int BM, int BN, int BK, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_kernel(const ConvParams p) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
    constexpr int LDA = BK + 4;
    constexpr int CH = BK / 4;            // float4 chunks per A row
    constexpr int RP = NT / CH;           // A rows covered per pass
    constexpr int A_IT = (BM + RP - 1) / RP;
    constexpr int B_F4 = BK * BN / 4;
    constexpr int B_IT = (B_F4 + NT - 1) / NT;
    static_assert(TM >= 1 && TN >= 1, "tile");
    static_assert(NT % CH == 0, "threads per row");

    __shared__ __attribute__((aligned(16))) float smem[2 * (BM * LDA + BK * BN)];
    float* sA0 = smem;
    float* sB0 = smem + 2 * BM * LDA;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int g = blockIdx.y;

    const int logical = xcd_remap(blockIdx.x, p.tilesM * p.tilesN);
    const int tile_m = logical / p.tilesN, tile_n = logical - tile_m * p.tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- per-thread A-row bookkeeping (fixed for the whole K loop)
    const int c4 = tid % CH;
    int a_by[A_IT], a_bx[A_IT], a_img[A_IT];
    bool a_ok[A_IT];
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
        a_img[ia] = img * p.H;
        a_by[ia] = oy * p.stride - p.pad;
        a_bx[ia] = ox * p.stride - p.pad;
        a_ok[ia] = ok;
    }

    const float* wg = p.w + (long long)g * p.wgroup_stride;

    // ---- K-loop state (wave-uniform)
    int ky = 0, kx = 0, s = 0, c0 = 0;
    const int KT = p.KH * p.KW * p.chunks_per_tap;

    f32x4 ra[A_IT], rb[B_IT];

    auto load_tile = [&](int kt) {
        const float* sp = p.src[s];
        const int ld = p.ld[s];
        const int cbase = p.coff[s] + g * p.cpg[s] + c0 + c4 * 4;
        const bool cok = (c0 + c4 * 4) < p.cpg[s];
#pragma unroll
        for (int ia = 0; ia < A_IT; ++ia) {
            const int iy = a_by[ia] + ky, ix = a_bx[ia] + kx;
            const bool ok = a_ok[ia] && cok && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(sp + ((long long)(a_img[ia] + iy) * p.W + ix) * ld + cbase);
            ra[ia] = v;
        }
#pragma unroll
        for (int ib = 0; ib < B_IT; ++ib) {
            const int f = tid + ib * NT;
            const int kq = f / BN, n = f - kq * BN;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((B_F4 % NT == 0 || f < B_F4) && (n0 + n) < p.Npad)
                v = *reinterpret_cast<const f32x4*>(wg + ((long long)(kt * (BK / 4) + kq) * p.Npad + n0 + n) * 4);
            rb[ib] = v;
        }
    };
    auto store_tile = [&](int buf) {
        float* sA = sA0 + buf * (BM * LDA);
        float* sB = sB0 + buf * (BK * BN);
#pragma unroll
        for (int ia = 0; ia < A_IT; ++ia) {
            const int row = tid / CH + ia * RP;
            if (row < BM) *reinterpret_cast<f32x4*>(sA + row * LDA + c4 * 4) = ra[ia];
        }
#pragma unroll
        for (int ib = 0; ib < B_IT; ++ib) {
            const int f = tid + ib * NT;
            if (B_F4 % NT == 0 || f < B_F4) *reinterpret_cast<f32x4*>(sB + f * 4) = rb[ib];
        }
    };
    auto advance = [&]() {
        c0 += BK;
        if (c0 >= p.cpg[s]) {
            c0 = 0;
            ++s;
            if (s == p.nsrc) {
                s = 0;
                ++kx;
                if (kx == p.KW) { kx = 0; ++ky; }
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = (kt + 1) < KT;
        if (more) {
            advance();
            load_tile(kt + 1);
        }
        mma_ktile<TM, TN, BK, LDA, BN>(sA0 + cur * (BM * LDA), sB0 + cur * (BK * BN), acc,
                                       wm * TM * 32, wn * TN * 32, lane);
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
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
                if (p.res) v += p.res[(long long)m * p.res_ld + p.res_coff + co];
                v = apply_act(v, p.act, p.slope);
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
    if (bk != 16 && bk != 32) return false;
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

template <int BM, int BN, int BK, int WGM, int WGN>
int launch_conv(ConvParams& p, int groups, hipStream_t st) {
    p.tilesM = cdiv(p.M, BM);
    p.tilesN = cdiv(p.Cout_g, BN);
    dim3 grid(p.tilesM * p.tilesN, groups, 1), block(64 * WGM * WGN, 1, 1);
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, BK, WGM, WGN>), grid, block, 0, st, p);
    E2_LAUNCH_CHECK("conv_igemm");
    return 0;
}

template <int BK>
int dispatch_tile(ConvParams& p, int groups, int tile, hipStream_t st) {
    switch (tile) {
        case 1: return launch_conv<128, 128, BK, 2, 2>(p, groups, st);
        case 2: return launch_conv<128, 64, BK, 2, 2>(p, groups, st);
        case 3: return launch_conv<64, 64, BK, 2, 2>(p, groups, st);
        case 4: return launch_conv<128, 32, BK, 4, 1>(p, groups, st);
        case 5: return launch_conv<64, 32, BK, 2, 1>(p, groups, st);
        case 6: return launch_conv<64, 128, BK, 2, 2>(p, groups, st);
        default: break;
    }
    e2fgvi_set_error("conv2d: unknown tile %d", tile);
    return E2FGVI_EINVAL;
}

int auto_tile(const ConvParams& p, int groups) {
    auto blocks = [&](int bm, int bn) { return (long long)cdiv(p.M, bm) * cdiv(p.Cout_g, bn) * groups; };
    const long long want = 2 * 256;    // >= 2 workgroups per CU before growing the tile
    if (p.Cout_g <= 32) return blocks(128, 32) >= want ? 4 : 5;
    if (p.Cout_g <= 64) return blocks(128, 64) >= want ? 2 : 3;
    if (blocks(128, 128) >= want) return 1;
    if (blocks(64, 128) >= want) return 6;
    return 3;
}

}  // namespace

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

extern "C" int e2fgvi_conv2d_nhwc(const e2fgvi_conv_desc* d, void* stream) {
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
    }
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
    if (!d->dst_nchw)
        E2_REQUIRE(d->dst_coff >= 0 && d->dst_coff + d->Cout <= d->dst_ld, E2FGVI_EINVAL, "conv2d: dst slice exceeds dst_ld");
    int tile = d->tile ? d->tile : auto_tile(p, d->groups);
    if (d->bk == 16) return dispatch_tile<16>(p, d->groups, tile, (hipStream_t)stream);
    return dispatch_tile<32>(p, d->groups, tile, (hipStream_t)stream);
}
