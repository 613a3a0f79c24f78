// Modulated deformable convolution (DCNv2) for gfx950 -- im2col-free.
//
// mmcv materialises the sampled columns [C*K, P] in HBM and calls a GEMM on them.  Here the
// sampled slab of one K-chunk (2 (group,tap) units x 16 channels for BM output pixels) is built
// straight in LDS by a bilinear gather and consumed by fp32 MFMA; the columns never exist.
//
// K order = (deform group g, tap, 16-channel sub-block, channel): with NHWC activations the 16
// channels of a unit are one contiguous 64-byte run per corner, so each corner fetch is four
// 16-byte lanes.  Offsets are data dependent, so the gather is software-pipelined two deep:
// offset/mask words of chunk k+2 and the 4 corner vectors of chunk k+1 are in flight while chunk
// k is multiplied.
//
// With `flows` set, SecondOrderDeformableAlignment's post-processing (feat_prop.py:38-53) is
// applied while loading: offset = max_residue*tanh(raw) + flow.flip, mask = sigmoid(raw).
//
// Replaces mmcv.ops.modulated_deform_conv2d at model/modules/feat_prop.py:55-58.
#include "common.h"

namespace {

struct DcnParams {
    const float* src[2];
    int ld[2];
    int c[2];
    int N, H, W, Ho, Wo, KH, KW, stride, pad, dil;
    int dg, cg, cgq;       // deform groups, channels per group, cg/16
    int KK;                // KH*KW
    int Cout, Npad;
    int M;
    int units;             // dg*KK*cgq
    const float* off; int off_ld;
    const float* msk; int msk_ld;
    const float* flows;
    float max_residue;
    const float* w;
    const float* bias;
    float* dst; int dst_ld, dst_coff;
    int tilesM, tilesN;
};

struct Sample {           // everything needed to fetch one (pixel, unit) x 4 channels
    float w00, w01, w10, w11;
    long long o00, o01, o10, o11;   // float offsets into the source (clamped, always valid)
    const float* base;
};

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(64 * WGM * WGN) void mdcn_kernel(const DcnParams p) {
    constexpr int BK = 32;
    constexpr int NT = 64 * WGM * WGN;
    constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
    constexpr int LDA = BK + 4;
    constexpr int A_ITEMS = BM * 8;                   // (row, unit-in-chunk, c4)
    constexpr int A_IT = (A_ITEMS + NT - 1) / NT;
    constexpr int B_F4 = BK * BN / 4;
    constexpr int B_IT = (B_F4 + NT - 1) / NT;

    __shared__ __attribute__((aligned(16))) float smem[2 * (BM * LDA + BK * BN)];
    float* sA0 = smem;
    float* sB0 = smem + 2 * BM * LDA;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int logical = xcd_remap(blockIdx.x, p.tilesM * p.tilesN);
    const int tile_m = logical / p.tilesN, tile_n = logical - tile_m * p.tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int HoWo = p.Ho * p.Wo;
    const int KT = (p.units + 1) / 2;

    // fixed per-thread item geometry
    int it_row[A_IT], it_uu[A_IT], it_c4[A_IT], it_img[A_IT], it_by[A_IT], it_bx[A_IT];
    long long it_pix[A_IT];
    bool it_ok[A_IT];
#pragma unroll
    for (int ia = 0; ia < A_IT; ++ia) {
        const int f = tid + ia * NT;
        const int row = f >> 3;
        it_row[ia] = row;
        it_uu[ia] = (f >> 2) & 1;
        it_c4[ia] = f & 3;
        const int m = m0 + row;
        const bool ok = (A_ITEMS % NT == 0 || f < A_ITEMS) && m < p.M;
        const int mm = ok ? m : 0;
        const int img = mm / HoWo;
        const int rem = mm - img * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        it_img[ia] = img;
        it_by[ia] = oy * p.stride - p.pad;
        it_bx[ia] = ox * p.stride - p.pad;
        it_pix[ia] = mm;
        it_ok[ia] = ok;
    }

    float r_dy[A_IT], r_dx[A_IT], r_mk[A_IT];      // raw offset words of the chunk after next
    f32x4 c00[A_IT], c01[A_IT], c10[A_IT], c11[A_IT];
    float w00[A_IT], w01[A_IT], w10[A_IT], w11[A_IT];
    f32x4 rb[B_IT];

    auto unit_of = [&](int kt, int uu, int& g, int& tap, int& cq) -> bool {
        const int u = kt * 2 + uu;
        if (u >= p.units) { g = 0; tap = 0; cq = 0; return false; }
        g = u / (p.KK * p.cgq);
        const int rem = u - g * (p.KK * p.cgq);
        tap = rem / p.cgq;
        cq = rem - tap * p.cgq;
        return true;
    };
    auto load_offsets = [&](int kt) {
#pragma unroll
        for (int ia = 0; ia < A_IT; ++ia) {
            int g, tap, cq;
            const bool uok = unit_of(kt, it_uu[ia], g, tap, cq);
            float dy = 0.f, dx = 0.f, mk = 0.f;
            if (uok && it_ok[ia]) {
                const float* po = p.off + it_pix[ia] * p.off_ld + (g * 2 * p.KK + 2 * tap);
                dy = po[0];
                dx = po[1];
                mk = p.msk[it_pix[ia] * p.msk_ld + g * p.KK + tap];
            }
            r_dy[ia] = dy; r_dx[ia] = dx; r_mk[ia] = mk;
        }
    };
    // turn the raw words (loaded for chunk kt) into corner fetches for chunk kt
    auto issue_corners = [&](int kt) {
#pragma unroll
        for (int ia = 0; ia < A_IT; ++ia) {
            int g, tap, cq;
            const bool uok = unit_of(kt, it_uu[ia], g, tap, cq);
            float dy = r_dy[ia], dx = r_dx[ia], mk = r_mk[ia];
            if (p.flows) {
                const float* fl = p.flows + it_pix[ia] * 4 + ((g * 2 >= p.dg) ? 2 : 0);
                const float fu = (uok && it_ok[ia]) ? fl[0] : 0.f, fv = (uok && it_ok[ia]) ? fl[1] : 0.f;
                dy = p.max_residue * tanhf(dy) + fv;     // flip: dy takes the v (y) component
                dx = p.max_residue * tanhf(dx) + fu;
                mk = 1.f / (1.f + expf(-mk));
            }
            const int ky = tap / p.KW, kx = tap - ky * p.KW;
            const float py = (float)(it_by[ia] + ky * p.dil) + dy;
            const float px = (float)(it_bx[ia] + kx * p.dil) + dx;
            const bool inside = uok && it_ok[ia] && py > -1.f && px > -1.f && py < (float)p.H && px < (float)p.W;
            const float fy = floorf(py), fx = floorf(px);
            const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
            const float ly = py - fy, lx = px - fx, hy = 1.f - ly, hx = 1.f - lx;
            const bool vy0 = y0 >= 0, vy1 = y1 <= p.H - 1, vx0 = x0 >= 0, vx1 = x1 <= p.W - 1;
            const float mm = inside ? mk : 0.f;
            w00[ia] = (vy0 && vx0) ? hy * hx * mm : 0.f;
            w01[ia] = (vy0 && vx1) ? hy * lx * mm : 0.f;
            w10[ia] = (vy1 && vx0) ? ly * hx * mm : 0.f;
            w11[ia] = (vy1 && vx1) ? ly * lx * mm : 0.f;
            const int cy0 = min(max(y0, 0), p.H - 1), cy1 = min(max(y1, 0), p.H - 1);
            const int cx0 = min(max(x0, 0), p.W - 1), cx1 = min(max(x1, 0), p.W - 1);
            int ch = g * p.cg + cq * 16 + it_c4[ia] * 4;
            const int s = (ch >= p.c[0]) ? 1 : 0;
            ch -= s ? p.c[0] : 0;
            const float* base = p.src[s] + ch;
            const long long ld = p.ld[s];
            const long long rowb = (long long)it_img[ia] * p.H;
            f32x4 z = {0.f, 0.f, 0.f, 0.f};
            if (inside) {
                c00[ia] = *reinterpret_cast<const f32x4*>(base + ((rowb + cy0) * p.W + cx0) * ld);
                c01[ia] = *reinterpret_cast<const f32x4*>(base + ((rowb + cy0) * p.W + cx1) * ld);
                c10[ia] = *reinterpret_cast<const f32x4*>(base + ((rowb + cy1) * p.W + cx0) * ld);
                c11[ia] = *reinterpret_cast<const f32x4*>(base + ((rowb + cy1) * p.W + cx1) * ld);
            } else {
                c00[ia] = z; c01[ia] = z; c10[ia] = z; c11[ia] = z;
            }
        }
    };
    auto load_w = [&](int kt) {
#pragma unroll
        for (int ib = 0; ib < B_IT; ++ib) {
            const int f = tid + ib * NT;
            const int kq = f / BN, n = f - kq * BN;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((B_F4 % NT == 0 || f < B_F4) && (n0 + n) < p.Npad)
                v = *reinterpret_cast<const f32x4*>(p.w + ((long long)(kt * (BK / 4) + kq) * p.Npad + n0 + n) * 4);
            rb[ib] = v;
        }
    };
    auto store_tile = [&](int buf) {
        float* sA = sA0 + buf * (BM * LDA);
        float* sB = sB0 + buf * (BK * BN);
#pragma unroll
        for (int ia = 0; ia < A_IT; ++ia) {
            if (A_ITEMS % NT == 0 || (tid + ia * NT) < A_ITEMS) {
                const f32x4 v = c00[ia] * w00[ia] + c01[ia] * w01[ia] + c10[ia] * w10[ia] + c11[ia] * w11[ia];
                *reinterpret_cast<f32x4*>(sA + it_row[ia] * LDA + it_uu[ia] * 16 + it_c4[ia] * 4) = v;
            }
        }
#pragma unroll
        for (int ib = 0; ib < B_IT; ++ib) {
            const int f = tid + ib * NT;
            if (B_F4 % NT == 0 || f < B_F4) *reinterpret_cast<f32x4*>(sB + f * 4) = rb[ib];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    // prologue: chunk 0 fully staged, offsets of chunk 1 in registers
    load_offsets(0);
    issue_corners(0);
    load_w(0);
    store_tile(0);
    if (KT > 1) load_offsets(1);
    __syncthreads();

    int cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = (kt + 1) < KT;
        if (more) {
            issue_corners(kt + 1);           // consumes r_* (offsets of chunk kt+1)
            load_w(kt + 1);
            if (kt + 2 < KT) load_offsets(kt + 2);
        }
        mma_ktile<TM, TN, BK, LDA, BN>(sA0 + cur * (BM * LDA), sB0 + cur * (BK * BN), acc,
                                       wm * TM * 32, wn * TN * 32, lane);
        if (more) store_tile(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + (wn * TN + tn) * 32 + j;
        if (n >= p.Cout) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int m = m0 + (wm * TM + tm) * 32 + row;
                if (m < p.M) p.dst[(long long)m * p.dst_ld + p.dst_coff + n] = acc[tm][tn][r] + bv;
            }
    }
}

__global__ void pack_dcn_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int C, int KK,
                                       int cg, int Npad, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = (int)(idx & 3);
    long long rem = idx >> 2;
    const int n = (int)(rem % Npad);
    const int kq = (int)(rem / Npad);
    const int k = kq * 4 + e;
    const int u = k >> 4, c = k & 15;
    const int cgq = cg / 16;
    const int g = u / (KK * cgq);
    const int r2 = u - g * (KK * cgq);
    const int tap = r2 / cgq, cq = r2 - tap * cgq;
    const int ch = g * cg + cq * 16 + c;
    float v = 0.f;
    if (ch < C && n < Cout && g * cg < C) v = w[((long long)n * C + ch) * KK + tap];
    wp[idx] = v;
}

long long dcn_packed_size(int Cout, int C, int KH, int KW) {
    const int units = (C / 16) * KH * KW;
    const int KT = (units + 1) / 2;
    return (long long)KT * 32 * round_up(Cout, 32);
}

template <int BM, int BN, int WGM, int WGN>
int launch_dcn(DcnParams& p, hipStream_t st) {
    p.tilesM = cdiv(p.M, BM);
    p.tilesN = cdiv(p.Cout, BN);
    hipLaunchKernelGGL((mdcn_kernel<BM, BN, WGM, WGN>), dim3(p.tilesM * p.tilesN), dim3(64 * WGM * WGN), 0, st, p);
    E2_LAUNCH_CHECK("mdcn");
    return 0;
}

}  // namespace

extern "C" int64_t e2fgvi_packed_dcn_weight_size(int32_t Cout, int32_t C, int32_t KH, int32_t KW) {
    if (Cout <= 0 || C <= 0 || C % 16 || KH <= 0 || KW <= 0) {
        e2fgvi_set_error("packed_dcn_weight_size: bad geometry");
        return E2FGVI_EINVAL;
    }
    return dcn_packed_size(Cout, C, KH, KW);
}

extern "C" int e2fgvi_pack_dcn_weight(const float* w, float* wpacked, int32_t Cout, int32_t C, int32_t KH, int32_t KW,
                                      int32_t deform_groups, void* stream) {
    E2_REQUIRE(w && wpacked, E2FGVI_EINVAL, "pack_dcn_weight: null pointer");
    E2_REQUIRE(Cout > 0 && C > 0 && deform_groups > 0 && C % deform_groups == 0 && (C / deform_groups) % 16 == 0,
               E2FGVI_EUNSUP, "pack_dcn_weight: channels per deform group must be a multiple of 16");
    const long long total = dcn_packed_size(Cout, C, KH, KW);
    hipLaunchKernelGGL(pack_dcn_weight_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       wpacked, Cout, C, KH * KW, C / deform_groups, round_up(Cout, 32), total);
    E2_LAUNCH_CHECK("pack_dcn_weight");
    return 0;
}

extern "C" int e2fgvi_mdcn_nhwc(const e2fgvi_mdcn_desc* d, void* stream) {
    E2_REQUIRE(d, E2FGVI_EINVAL, "mdcn: null descriptor");
    E2_REQUIRE(d->nsrc == 1 || d->nsrc == 2, E2FGVI_EINVAL, "mdcn: nsrc must be 1 or 2");
    DcnParams p;
    int C = 0;
    for (int s = 0; s < 2; ++s) { p.src[s] = nullptr; p.ld[s] = 0; p.c[s] = 0; }
    for (int s = 0; s < d->nsrc; ++s) {
        E2_REQUIRE(d->src[s] && d->src_c[s] > 0 && d->src_ld[s] >= d->src_c[s] && d->src_ld[s] % 4 == 0 &&
                       ((uintptr_t)d->src[s] & 15) == 0,
                   E2FGVI_EINVAL, "mdcn: bad source %d", s);
        p.src[s] = d->src[s]; p.ld[s] = d->src_ld[s]; p.c[s] = d->src_c[s];
        C += d->src_c[s];
    }
    if (d->nsrc == 1) { p.src[1] = p.src[0]; p.ld[1] = p.ld[0]; p.c[1] = 0; }
    E2_REQUIRE(d->deform_groups > 0 && C % d->deform_groups == 0, E2FGVI_EINVAL, "mdcn: C %% deform_groups != 0");
    const int cg = C / d->deform_groups;
    E2_REQUIRE(cg % 16 == 0, E2FGVI_EUNSUP, "mdcn: channels per deform group (%d) must be a multiple of 16", cg);
    E2_REQUIRE(d->nsrc == 1 || d->src_c[0] % cg == 0, E2FGVI_EUNSUP, "mdcn: a deform group straddles the two sources");
    E2_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->dil > 0 && d->pad >= 0 && d->N > 0, E2FGVI_EINVAL, "mdcn: bad sizes");
    E2_REQUIRE(d->Ho == (d->H + 2 * d->pad - (d->dil * (d->KH - 1) + 1)) / d->stride + 1 &&
                   d->Wo == (d->W + 2 * d->pad - (d->dil * (d->KW - 1) + 1)) / d->stride + 1,
               E2FGVI_EINVAL, "mdcn: Ho/Wo inconsistent");
    E2_REQUIRE(d->offset && d->mask && d->wpacked && d->dst, E2FGVI_EINVAL, "mdcn: null pointer");
    E2_REQUIRE(!d->flows || d->deform_groups % 2 == 0, E2FGVI_EINVAL, "mdcn: fused flows need an even group count");
    E2_REQUIRE(d->dst_coff + d->Cout <= d->dst_ld, E2FGVI_EINVAL, "mdcn: dst slice exceeds dst_ld");
    p.N = d->N; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad; p.dil = d->dil;
    p.dg = d->deform_groups; p.cg = cg; p.cgq = cg / 16; p.KK = d->KH * d->KW;
    p.Cout = d->Cout; p.Npad = round_up(d->Cout, 32);
    p.M = d->N * d->Ho * d->Wo;
    p.units = p.dg * p.KK * p.cgq;
    p.off = d->offset; p.off_ld = d->off_ld; p.msk = d->mask; p.msk_ld = d->mask_ld;
    p.flows = d->flows; p.max_residue = d->max_residue;
    p.w = d->wpacked; p.bias = d->bias;
    p.dst = d->dst; p.dst_ld = d->dst_ld; p.dst_coff = d->dst_coff;
    int tile = d->tile;
    if (!tile) {
        const long long b64 = (long long)cdiv(p.M, 64) * cdiv(p.Cout, 128);
        tile = b64 >= 512 ? 1 : 2;
    }
    if (tile == 1) return launch_dcn<64, 128, 2, 2>(p, (hipStream_t)stream);
    if (tile == 2) return launch_dcn<32, 128, 1, 4>(p, (hipStream_t)stream);
    if (tile == 3) return launch_dcn<32, 64, 1, 2>(p, (hipStream_t)stream);
    e2fgvi_set_error("mdcn: unknown tile %d", tile);
    return E2FGVI_EINVAL;
}
