// bf16-MFMA variant of the implicit-GEMM NHWC convolution / linear layer (gfx950).
//
// Optional precision mode for the HQ configurations (BASELINE.json configs 3/4: "bf16 MFMA"): activations stay fp32
// in HBM, are rounded to bf16 (RNE, v_cvt_pk_bf16_f32) on their way into LDS, weights are pre-packed as bf16, products
// run on v_mfma_f32_32x32x16_bf16 (16x the fp32 MFMA rate) and accumulate in fp32; the epilogue (bias, residual,
// activation, fp32 store) is the fp32 kernel's.  The default path of the library is fp32 -- this kernel is only used
// when the caller asks for it (Engine(precision="bf16")).
//
// ROUND 3: no longer part of the product library (the bf16 data path is conv_bf16x.hip).  Kept here as the AGGRESSOR of
// the cross-stream reproducer (overlap_probe.hip): its tiles with 2x2 MFMA accumulators are the ones beside which
// packed-fp32 VALU on VMEM-fresh registers misbehaves.  Build:
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Ie2fgvi_amd/csrc -Iinclude tools/probe/conv_bf16_r1.hip \
//           e2fgvi_amd/csrc/error.hip -o /tmp/libe2fgvi_r1_aggressor.so
//
// Structure: same as conv.hip's conv_igemm_kernel (buffer loads with hardware zeroing, register prefetch stages,
// 2-deep LDS ring, XCD-aware tiles), with a 64-deep K-step = two 32-channel chunks per barrier.
//   operands of v_mfma_f32_32x32x16_bf16: lane l holds 8 consecutive k of row/col (l & 31), k-block (l >> 5)
//   LDS A: [BM][64 (+8 pad)] bf16 (144-byte rows: conflict-free 16-byte reads), LDS B: [8 k-octets][BN][8] bf16
//   packed weights: [group][K/8][Npad][8] bf16, K = tap-major, sources padded to 32 channels
#include "common.h"

extern "C" {
int64_t e2fgvi_packed_conv_weight_bf16_size(int32_t Cout, int32_t groups, int32_t KH, int32_t KW,
                                            int32_t nsrc, const int32_t* src_cpg);   /* in bf16 elements */
int e2fgvi_pack_conv_weight_bf16(const float* w, void* wpacked, int32_t Cout, int32_t groups, int32_t KH,
                                 int32_t KW, int32_t nsrc, const int32_t* src_cpg, void* stream);
int e2fgvi_conv2d_nhwc_bf16(const e2fgvi_conv_desc* d, void* stream);
}

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ConvParamsB {
    const float* src[E2FGVI_MAX_SRC];
    int ld[E2FGVI_MAX_SRC];
    int coff[E2FGVI_MAX_SRC];
    int cpg[E2FGVI_MAX_SRC];
    int nsrc;
    int N, H, W, Ho, Wo, KH, KW, stride, pad;
    int Cout, Cout_g, Npad;
    int M;
    int tilesM, tilesN;
    int chunks_per_tap;                   // 32-channel chunks per tap
    unsigned src_bytes[E2FGVI_MAX_SRC];
    unsigned wgroup_bytes;
    long long wgroup_elems;               // bf16 elements per group
    const __bf16* w;
    const float* bias;
    const float* res;
    int res_ld, res_coff;
    float* dst;
    int dst_ld, dst_coff, dst_nchw;
    int act;
    float slope;
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ u32x4 buf_load4u(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
}

template <int BM, int BN, int WGM, int WGN, int D>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_igemm_bf16_kernel(const ConvParamsB p) {
    constexpr int NT = 64 * WGM * WGN;
    constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
    constexpr int LDA = 72;                              // bf16 elements per A row (64 + 8 pad)
    constexpr int A_F4 = BM * 8;                         // fp32 float4s of ONE 32-channel sub-chunk
    constexpr int A_IT = (A_F4 + NT - 1) / NT;
    constexpr int B_V = 8 * BN;                          // 16-byte entries (8 bf16) of the weight slab of one K-step
    constexpr int B_IT = (B_V + NT - 1) / NT;
    constexpr int STAGE_B = BM * LDA * 2 + B_V * 16;     // bytes of one LDS stage
    constexpr unsigned OOB = 0xFFFFFFFFu;
    static_assert(TM >= 1 && TN >= 1, "tile");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE_B];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int g = blockIdx.y;
    const int logical = xcd_remap(blockIdx.x, p.tilesM * p.tilesN);
    const int tile_m = logical / p.tilesN, tile_n = logical - tile_m * p.tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // per-thread A-row bookkeeping
    int a_pix[A_IT], a_by[A_IT], a_bx[A_IT];
    const int HoWo = p.Ho * p.Wo;
#pragma unroll
    for (int ia = 0; ia < A_IT; ++ia) {
        const int f = tid + ia * NT;
        const int row = f >> 3;
        const int m = m0 + row;
        const bool ok = (A_F4 % NT == 0 || f < A_F4) && (m < p.M);
        const int mm = ok ? m : 0;
        const int img = mm / HoWo;
        const int rem = mm - img * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        a_by[ia] = ok ? oy * p.stride - p.pad : -(1 << 28);
        a_bx[ia] = ox * p.stride - p.pad;
        a_pix[ia] = (img * p.H + oy * p.stride - p.pad) * p.W + ox * p.stride - p.pad;
    }
    const int c4 = tid & 7;
    unsigned b_off[B_IT];
#pragma unroll
    for (int ib = 0; ib < B_IT; ++ib) {
        const int f = tid + ib * NT;
        const int oct = f / BN, n = f - oct * BN;
        const bool ok = (B_V % NT == 0 || f < B_V) && (n0 + n) < p.Npad;
        b_off[ib] = ok ? (unsigned)((oct * p.Npad + n0 + n) * 16) : OOB;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(p.w + (long long)g * p.wgroup_elems, p.wgroup_bytes);
    const unsigned b_step = 8u * (unsigned)p.Npad * 16u;          // bytes per 64-deep K-step

    int ky = 0, kx = 0, s = 0, c0 = 0;                             // next 32-channel chunk to load
    const int KT32 = p.KH * p.KW * p.chunks_per_tap;
    const int nStep = (KT32 + 1) / 2;

    f32x4 ra[D][2][A_IT];
    u32x4 rb[D][B_IT];

    // current source's parameters in scalar registers (see conv.hip::conv_igemm_kernel)
    const float* cur_src = p.src[0];
    unsigned cur_bytes = p.src_bytes[0];
    unsigned cur_ld4 = (unsigned)p.ld[0] * 4u;
    unsigned cur_chan = (unsigned)(p.coff[0] + g * p.cpg[0]) * 4u;
    int cur_cpg = p.cpg[0];
    auto advance = [&]() {
        c0 += 32;
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
    auto load_step = [&](int step, f32x4 (&qa)[2][A_IT], u32x4 (&qb)[B_IT]) {
#pragma unroll
        for (int sc = 0; sc < 2; ++sc) {
            const bool chunk_ok = (2 * step + sc) < KT32;
            const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(cur_src, cur_bytes);
            const unsigned ld4 = cur_ld4;
            const unsigned chan = cur_chan + (unsigned)(c0 + c4 * 4) * 4u;
            const bool cok = chunk_ok && (c0 + c4 * 4) < cur_cpg;
            const int tap = ky * p.W + kx;
#pragma unroll
            for (int ia = 0; ia < A_IT; ++ia) {
                const bool ok = cok && (unsigned)(a_by[ia] + ky) < (unsigned)p.H && (unsigned)(a_bx[ia] + kx) < (unsigned)p.W;
                const unsigned off = (unsigned)(a_pix[ia] + tap) * ld4 + chan;
                qa[sc][ia] = buf_load4(arsrc, ok ? off : OOB);
            }
            advance();
        }
        const bool step_ok = step < nStep;
#pragma unroll
        for (int ib = 0; ib < B_IT; ++ib)
            qb[ib] = buf_load4u(wrsrc, (b_off[ib] == OOB || !step_ok) ? OOB : b_off[ib] + (unsigned)step * b_step);
    };
    auto store_step = [&](int buf, const f32x4 (&qa)[2][A_IT], const u32x4 (&qb)[B_IT]) {
        __bf16* sA = reinterpret_cast<__bf16*>(smem + buf * STAGE_B);
        unsigned char* sB = smem + buf * STAGE_B + BM * LDA * 2;
#pragma unroll
        for (int sc = 0; sc < 2; ++sc)
#pragma unroll
            for (int ia = 0; ia < A_IT; ++ia) {
                const int f = tid + ia * NT;
                if (A_F4 % NT == 0 || f < A_F4) {
                    const f32x4 v = qa[sc][ia];
                    bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *reinterpret_cast<bf16x4*>(sA + (f >> 3) * LDA + sc * 32 + c4 * 4) = h;
                }
            }
#pragma unroll
        for (int ib = 0; ib < B_IT; ++ib) {
            const int f = tid + ib * NT;
            if (B_V % NT == 0 || f < B_V) *reinterpret_cast<u32x4*>(sB + f * 16) = qb[ib];
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int nIter = (nStep + D - 1) / D * D;
    load_step(0, ra[0], rb[0]);
    store_step(0, ra[0], rb[0]);
#pragma unroll
    for (int j = 1; j < D; ++j) load_step(j, ra[j], rb[j]);
    __syncthreads();

    const int i = lane & 31, h = lane >> 5;
    int cur = 0;
    for (int it0 = 0; it0 < nIter; it0 += D) {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const int step = it0 + j;
            load_step(step + D, ra[j], rb[j]);
            const __bf16* sA = reinterpret_cast<const __bf16*>(smem + cur * STAGE_B);
            const __bf16* sB = reinterpret_cast<const __bf16*>(smem + cur * STAGE_B + BM * LDA * 2);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    a[tm] = *reinterpret_cast<const bf16x8*>(sA + ((wm * TM + tm) * 32 + i) * LDA + st * 16 + h * 8);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    b[tn] = *reinterpret_cast<const bf16x8*>(sB + ((2 * st + h) * BN + (wn * TN + tn) * 32 + i) * 8);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
            }
            store_step(cur ^ 1, ra[(j + 1) % D], rb[(j + 1) % D]);
            __syncthreads();
            cur ^= 1;
        }
    }

    // ---- epilogue (fp32)
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + (wn * TN + tn) * 32 + i;
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

struct PackB {
    int Cout, groups, KH, KW, nsrc;
    int cpg[E2FGVI_MAX_SRC];
    int Cout_g, Npad, Cin_g, chunks_per_tap;
    long long total;            // bf16 elements
    long long wgroup_elems;
};

bool geometry_b(int Cout, int groups, int KH, int KW, int nsrc, const int32_t* cpg, PackB* q) {
    if (Cout <= 0 || groups <= 0 || Cout % groups || KH <= 0 || KW <= 0 || nsrc < 1 || nsrc > E2FGVI_MAX_SRC) return false;
    q->Cout = Cout; q->groups = groups; q->KH = KH; q->KW = KW; q->nsrc = nsrc;
    q->Cout_g = Cout / groups;
    q->Npad = round_up(q->Cout_g, 32);
    q->Cin_g = 0;
    q->chunks_per_tap = 0;
    for (int s = 0; s < E2FGVI_MAX_SRC; ++s) q->cpg[s] = 0;
    for (int s = 0; s < nsrc; ++s) {
        if (cpg[s] <= 0 || cpg[s] % 4) return false;
        q->cpg[s] = cpg[s];
        q->Cin_g += cpg[s];
        q->chunks_per_tap += cdiv(cpg[s], 32);
    }
    // K is padded to a whole number of 64-deep steps so the last step's second half reads zeros inside the buffer
    const long long kchunks = (long long)KH * KW * q->chunks_per_tap;
    q->wgroup_elems = (kchunks + (kchunks & 1)) * 32 * q->Npad;
    q->total = q->wgroup_elems * groups;
    return true;
}

__global__ void pack_conv_weight_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, const PackB p) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.total) return;
    const int g = (int)(idx / p.wgroup_elems);
    long long rem = idx - (long long)g * p.wgroup_elems;
    const int e = (int)(rem & 7);
    rem >>= 3;
    const int n = (int)(rem % p.Npad);
    const int ko = (int)(rem / p.Npad);
    const int k = ko * 8 + e;
    const int kt = k / 32, kk = k - kt * 32;
    const int tap = kt / p.chunks_per_tap;
    float v = 0.f;
    if (tap < p.KH * p.KW) {
        int chunk = kt - tap * p.chunks_per_tap;
        int s = 0, prefix = 0;
        while (true) {
            const int nc = (p.cpg[s] + 31) / 32;
            if (chunk < nc) break;
            chunk -= nc;
            prefix += p.cpg[s];
            ++s;
        }
        const int c = chunk * 32 + kk;
        if (c < p.cpg[s] && n < p.Cout_g)
            v = w[((long long)(g * p.Cout_g + n) * p.Cin_g + prefix + c) * (p.KH * p.KW) + tap];
    }
    wp[idx] = (__bf16)v;
}

template <int BM, int BN, int WGM, int WGN, int D>
int launch_b(ConvParamsB& p, int groups, hipStream_t st) {
    p.tilesM = cdiv(p.M, BM);
    p.tilesN = cdiv(p.Cout_g, BN);
    hipLaunchKernelGGL((conv_igemm_bf16_kernel<BM, BN, WGM, WGN, D>), dim3(p.tilesM * p.tilesN, groups, 1),
                       dim3(64 * WGM * WGN), 0, st, p);
    E2_LAUNCH_CHECK("conv_igemm_bf16");
    return 0;
}

}  // namespace

extern "C" int64_t e2fgvi_packed_conv_weight_bf16_size(int32_t Cout, int32_t groups, int32_t KH, int32_t KW, int32_t nsrc,
                                                       const int32_t* src_cpg) {
    PackB q;
    if (!src_cpg || !geometry_b(Cout, groups, KH, KW, nsrc, src_cpg, &q)) {
        e2fgvi_set_error("packed_conv_weight_bf16_size: bad geometry");
        return E2FGVI_EINVAL;
    }
    return q.total;
}

extern "C" int e2fgvi_pack_conv_weight_bf16(const float* w, void* wpacked, int32_t Cout, int32_t groups, int32_t KH,
                                            int32_t KW, int32_t nsrc, const int32_t* src_cpg, void* stream) {
    PackB q;
    E2_REQUIRE(w && wpacked && src_cpg, E2FGVI_EINVAL, "pack_conv_weight_bf16: null pointer");
    E2_REQUIRE(geometry_b(Cout, groups, KH, KW, nsrc, src_cpg, &q), E2FGVI_EINVAL, "pack_conv_weight_bf16: bad geometry");
    hipLaunchKernelGGL(pack_conv_weight_bf16_kernel, dim3((unsigned)cdiv64(q.total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, (__bf16*)wpacked, q);
    E2_LAUNCH_CHECK("pack_conv_weight_bf16");
    return 0;
}

extern "C" int e2fgvi_conv2d_nhwc_bf16(const e2fgvi_conv_desc* d, void* stream) {
    E2_REQUIRE(d, E2FGVI_EINVAL, "conv2d_bf16: null descriptor");
    PackB q;
    E2_REQUIRE(geometry_b(d->Cout, d->groups, d->KH, d->KW, d->nsrc, d->src_cpg, &q), E2FGVI_EINVAL, "conv2d_bf16: bad geometry");
    E2_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->stride > 0 && d->pad >= 0, E2FGVI_EINVAL,
               "conv2d_bf16: bad sizes");
    E2_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1,
               E2FGVI_EINVAL, "conv2d_bf16: Ho/Wo inconsistent with H/W/k/stride/pad");
    E2_REQUIRE((long long)d->N * d->Ho * d->Wo < 2147483647LL, E2FGVI_EUNSUP, "conv2d_bf16: more than 2^31 output pixels");
    E2_REQUIRE(d->wpacked && d->dst, E2FGVI_EINVAL, "conv2d_bf16: null weight/dst");
    E2_REQUIRE(d->act != E2FGVI_ACT_DCNPOST, E2FGVI_EUNSUP, "conv2d_bf16: ACT_DCNPOST is fp32-only");
    ConvParamsB p;
    for (int s = 0; s < E2FGVI_MAX_SRC; ++s) { p.src[s] = nullptr; p.ld[s] = 0; p.coff[s] = 0; p.cpg[s] = 0; p.src_bytes[s] = 0; }
    for (int s = 0; s < d->nsrc; ++s) {
        E2_REQUIRE(d->src[s], E2FGVI_EINVAL, "conv2d_bf16: null source %d", s);
        E2_REQUIRE(d->src_ld[s] % 4 == 0 && d->src_coff[s] % 4 == 0 && ((uintptr_t)d->src[s] & 15) == 0, E2FGVI_EINVAL,
                   "conv2d_bf16: source %d not 16-byte addressable", s);
        E2_REQUIRE(d->src_coff[s] + d->groups * d->src_cpg[s] <= d->src_ld[s], E2FGVI_EINVAL,
                   "conv2d_bf16: source %d channel range exceeds its pixel stride", s);
        const long long bytes = (long long)d->N * d->H * d->W * d->src_ld[s] * 4;
        E2_REQUIRE(bytes < 4294967295LL, E2FGVI_EUNSUP, "conv2d_bf16: source %d spans >= 4 GiB (split the batch)", s);
        p.src[s] = d->src[s]; p.ld[s] = d->src_ld[s]; p.coff[s] = d->src_coff[s]; p.cpg[s] = d->src_cpg[s];
        p.src_bytes[s] = (unsigned)bytes;
    }
    E2_REQUIRE(q.wgroup_elems * 2 < 4294967295LL, E2FGVI_EUNSUP, "conv2d_bf16: packed weight group >= 4 GiB");
    E2_REQUIRE(((uintptr_t)d->wpacked & 15) == 0, E2FGVI_EINVAL, "conv2d_bf16: packed weight not 16-byte aligned");
    if (!d->dst_nchw)
        E2_REQUIRE(d->dst_coff >= 0 && d->dst_coff + d->Cout <= d->dst_ld, E2FGVI_EINVAL, "conv2d_bf16: dst slice exceeds dst_ld");
    p.nsrc = d->nsrc;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
    p.Cout = d->Cout; p.Cout_g = q.Cout_g; p.Npad = q.Npad;
    p.M = d->N * d->Ho * d->Wo;
    p.chunks_per_tap = q.chunks_per_tap;
    p.wgroup_elems = q.wgroup_elems; p.wgroup_bytes = (unsigned)(q.wgroup_elems * 2);
    p.w = (const __bf16*)d->wpacked; p.bias = d->bias; p.res = d->residual; p.res_ld = d->res_ld; p.res_coff = d->res_coff;
    p.dst = d->dst; p.dst_ld = d->dst_ld; p.dst_coff = d->dst_coff; p.dst_nchw = d->dst_nchw;
    p.act = d->act; p.slope = d->slope;
    hipStream_t st = (hipStream_t)stream;
    int tile = d->tile;
    if (!tile) {
        auto blocks = [&](int bm, int bn) { return (long long)cdiv(p.M, bm) * cdiv(p.Cout_g, bn) * d->groups; };
        // measured (tools/bf16_bench.py): the kernel is load-bound, one register stage and 128x128 tiles are best
        if (p.Cout_g <= 32) tile = 4;
        else if (p.Cout_g <= 64) tile = 2;
        else tile = blocks(128, 128) >= 512 ? 5 : 3;
    }
    switch (tile) {
        case 1: return launch_b<128, 128, 2, 2, 2>(p, d->groups, st);
        case 2: return launch_b<128, 64, 2, 2, 2>(p, d->groups, st);
        case 3: return launch_b<64, 64, 2, 2, 2>(p, d->groups, st);
        case 4: return launch_b<128, 32, 4, 1, 2>(p, d->groups, st);
        case 5: return launch_b<128, 128, 2, 2, 1>(p, d->groups, st);
        case 6: return launch_b<256, 128, 4, 2, 2>(p, d->groups, st);
        default: break;
    }
    e2fgvi_set_error("conv2d_bf16: unknown tile %d", tile);
    return E2FGVI_EINVAL;
}
