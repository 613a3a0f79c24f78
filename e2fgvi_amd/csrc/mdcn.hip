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
#include <stdlib.h>

namespace {

constexpr int MAX_UNITS = 320;      // (C / 16) * KH * KW entries of the unit table (144 for the 256-channel 3x3 DCN)

struct DcnParams {
    const void* src[2];                // fp32 NHWC, or bf16 NHWC in the S16 instantiation
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
    float* dst; int dst_ld, dst_coff, dst_bf16;
    int tilesM, tilesN;
    int planar;            // bf16 sources laid out [group][pixel][16]: plane_bytes per group
    unsigned plane_bytes;
    int sw, swX, swY;      // sw: a tile's BM rows are an 8 x (BM / 8) pixel block (swX x swY blocks per image) instead of BM consecutive pixels
    int units0;            // units (of 16 channels x tap) that live in source 0
    unsigned src_bytes[2], off_bytes, msk_bytes, flw_bytes, w_bytes;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 buf_load2(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ float buf_load1(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)byte_off, 0, 0));
}

__device__ __forceinline__ float fast_sigmoid(float x) {
    return __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float fast_tanh(float x) {
    // 1 - 2/(1 + e^{2x}); saturates correctly for large |x| (exp2 -> inf / 0)
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(2.8853900817779268f * x));
}

// KS K-groups of WGM x WGN waves share one output tile (intra-workgroup split-K, see conv.hip): group kg takes the
// K-chunks kg, kg+KS, ... .  All loads are raw buffer loads: corners outside the image, rows outside the problem and
// chunks past the end read as zero through the buffer bounds, so the loop has no branches.
// BF: the sampled slab and the weights are held in LDS as bf16 and multiplied on v_mfma_f32_32x32x16_bf16 (the bf16 data
// path); the gather, the bilinear blend and the accumulation are fp32 either way.
//   bf16 LDS image: A [BM rows][32 k (+8 pad)] (80-byte rows: conflict-free b128 reads).  The B operand (weights) does NOT go
//   through LDS (round 5): the packed layout [chunk][plane][4 k-octets][Npad][8] is exactly the MFMA's B operand per lane (lane
//   (n = l & 31, half = l >> 5) of k-step kk needs the 16 bytes of octet 2 kk + half, column n), so every wave fetches the
//   fragments of its own columns straight into registers, a chunk ahead -- no staging registers -> ds_write -> ds_read round trip
//   (6 + 6 LDS instructions per thread and chunk in the split-operand variant), and an LDS stage shrinks to the A slab
// S16 (with BF): the sources are bf16 NHWC.  A 16-byte corner fetch then carries 8 channels instead of 4, so an item is
// (row, unit, 8-channel half): half the items, half the fetches and half the per-item offset arithmetic (the bilinear
// weights, tanh / sigmoid and address math are computed once per item) for the same slab.
// X3 (with BF, fp32 sources; round 3): the fp32 layer on the bf16 matrix pipe -- every blended value is split EXACTLY into three
// bf16 numbers (hi = the value with its low 16 bits cleared, mid = the same of the exact remainder, lo = the rest) by the
// thread that blends it, once per workgroup, the weights are packed as three bf16 planes whose sum is the fp32 weight, and six
// of the nine bf16 partial products are accumulated (conv_bf16x.hip MODE 2): fp32-level rounding at 6 x 32 instead of 8 x 64
// MFMA cycles per 16 k.  LDS images: A [3 planes][BM rows][32 k (+8 pad)], B [3 planes][4 k-octets][BN][8].
// Static LDS of one mdcn_kernel instantiation in bytes: the ring (or the K groups' partial sums that reuse it, whichever is larger),
// the unit table and the raw offset / mask slots.  The kernel sizes its arrays and launch_dcn() decides "fits / EUNSUP" from this
// ONE expression (round 6, advisor: the guards counted the ring only while the kernel took max(ring, partial sums)).
template <int BM, int BN, int WGM, int WGN, int KS, bool BF, bool X3>
constexpr int mdcn_smem_floats() {
    constexpr int NG = 64 * WGM * WGN, TM = BM / (32 * WGM), TN = BN / (32 * WGN);
    constexpr int STAGE = BF ? (X3 ? 3 : 1) * (BM * (32 + 8)) / 2 : BM * (32 + 4) + 32 * BN;
    constexpr int RING = KS * 2 * STAGE, PARTS = (KS - 1) * NG * TM * TN * 16;
    return RING > PARTS ? RING : PARTS;
}
template <int BM, int BN, int WGM, int WGN, int KS, bool BF, bool X3>
constexpr bool mdcn_fits_lds() {
    return mdcn_smem_floats<BM, BN, WGM, WGN, KS, BF, X3>() * 4 + MAX_UNITS * 8 * 4 + KS * 2 * (2 * BM) * 8 * 4 + 1024 <= 160 * 1024;
}

template <int BM, int BN, int WGM, int WGN, int KS, bool BF, bool S16, bool X3 = false>
__global__ __launch_bounds__(64 * WGM * WGN * KS) void mdcn_kernel(const DcnParams p) {
    static_assert(BF || !S16, "bf16 sources only with the bf16 MFMA slab");
    static_assert(!X3 || (BF && !S16), "the split-operand variant: bf16 LDS images of fp32 sources");
    constexpr int NP = X3 ? 3 : 1;                    // bf16 planes per operand
    constexpr int SB = S16 ? 2 : 4;                   // source element size
    constexpr int CQ = S16 ? 2 : 4;                   // 16-byte corner fetches per (row, unit): 16 channels
    constexpr int BK = 32;
    constexpr int NG = 64 * WGM * WGN;
    constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
    constexpr int LDA = BK + 4;
    constexpr int LDA16 = BK + 8;                     // bf16 elements per A row
    constexpr int A_ITEMS = BM * 2 * CQ;              // (row, unit-in-chunk, 16-byte channel part)
    constexpr int A_IT = (A_ITEMS + NG - 1) / NG;
    constexpr int B_F4 = BK * BN / 4;                 // fp32 MFMA path: 16-byte items of a chunk's weight slab (staged through LDS)
    constexpr int B_IT = BF ? 1 : (B_F4 + NG - 1) / NG;
    constexpr int STAGE = BF ? NP * (BM * LDA16) / 2 : BM * LDA + BK * BN;      // floats (bf16 products: the A slab only)
    constexpr unsigned OOB = 0xFFFFFFFFu;
    constexpr int SMEM_F = mdcn_smem_floats<BM, BN, WGM, WGN, KS, BF, X3>();     // ring; the K groups' partial sums reuse it
    static_assert(SMEM_F >= KS * 2 * STAGE, "mdcn_smem_floats() and the kernel's stage layout disagree");

    __shared__ __attribute__((aligned(16))) float smem[SMEM_F];
    __shared__ __attribute__((aligned(16))) int utab[MAX_UNITS * 8];
    // raw offset / mask (/ flow) words of a chunk: one 32-byte slot per (pixel row, unit) = {dy, dx, mask, -, fu, fv, -, -},
    // fetched by the first 2 BM threads of the group (3 loads) instead of by every item's four channel lanes
    // (20 loads per wave and chunk, mostly distinct cache lines); two buffers per group
    constexpr int SLOTS = 2 * BM;
    static_assert(SLOTS % 64 == 0 && SLOTS <= NG, "loader threads are whole waves of the group");
    __shared__ __attribute__((aligned(16))) float sraw[KS * 2 * SLOTS * 8];

    const int kg = (KS == 1) ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x / NG));
    const int tid = threadIdx.x - kg * NG;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    float* sbase = smem + kg * (2 * STAGE);
    const int logical = xcd_remap(blockIdx.x, p.tilesM * p.tilesN);
    const int tile_m = logical / p.tilesN, tile_n = logical - tile_m * p.tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int HoWo = p.Ho * p.Wo;
    const int KT = (p.units + 1) / 2;

    const __amdgpu_buffer_rsrc_t r_src0 = make_rsrc(p.src[0], p.src_bytes[0]);
    const __amdgpu_buffer_rsrc_t r_src1 = make_rsrc(p.src[1], p.src_bytes[1]);
    const __amdgpu_buffer_rsrc_t r_off = make_rsrc(p.off, p.off_bytes);
    const __amdgpu_buffer_rsrc_t r_msk = make_rsrc(p.msk, p.msk_bytes);
    const __amdgpu_buffer_rsrc_t r_flw = make_rsrc(p.flows ? p.flows : p.off, p.flows ? p.flw_bytes : 0u);
    const __amdgpu_buffer_rsrc_t r_w = make_rsrc(p.w, p.w_bytes);

    // fixed per-thread item geometry
    int it_row[A_IT], it_uu[A_IT], it_c4[A_IT];
    bool it_ok[A_IT];
#pragma unroll
    for (int ia = 0; ia < A_IT; ++ia) {
        const int f = tid + ia * NG;
        it_row[ia] = f / (2 * CQ);
        it_uu[ia] = (f / CQ) & 1;
        it_c4[ia] = f & (CQ - 1);
        it_ok[ia] = A_ITEMS % NG == 0 || f < A_ITEMS;
    }
    unsigned b_off[B_IT];
#pragma unroll
    for (int ib = 0; ib < B_IT; ++ib) {
        const int f = tid + ib * NG;
        const int kq = f / BN, n = f - kq * BN;
        const bool ok = (B_F4 % NG == 0 || f < B_F4) && (n0 + n) < p.Npad;
        b_off[ib] = ok ? (unsigned)((kq * p.Npad + n0 + n) * 16) : OOB;
    }
    const unsigned b_step = (unsigned)(BF ? NP * BK / 8 : BK / 4) * (unsigned)p.Npad * 16u;
    // bf16 products: the lane's B fragments of chunk kt = 16 bytes at (plane pl, octet 2 kk + half, column n0 + (wn TN + tn) 32 + n).
    // One lane-dependent offset (half, n) advanced by a chunk per trip; plane / k-step / column tile are wave-uniform and go into
    // the load's scalar offset.  A column tile that lies entirely past Npad (Cout < BN) re-reads tile 0 -- its results are never
    // stored, and the scalar offset (which the buffer's range check does not see) must not lead outside the packed weights.
    const unsigned bf_lane = (unsigned)(((lane >> 5) * p.Npad + n0 + (lane & 31)) * 16);
    unsigned bf_tn[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn)
        bf_tn[tn] = (unsigned)__builtin_amdgcn_readfirstlane((n0 + (wn * TN + tn) * 32 < p.Npad) ? (wn * TN + tn) * 32 * 16 : 0);
    // (readfirstlane: wn is wave-uniform but derived from threadIdx; as a vector value the scalar offset made hipcc wrap every
    //  fragment load in a waterfall loop)
    const unsigned bf_oct2 = 2u * (unsigned)p.Npad * 16u;       // two k-octets (one k-step of 16)

    // two register sets: the corner fetches / weights of chunk k+2 are issued while those of chunk k+1 (issued one
    // iteration earlier) are blended into LDS -- a whole MFMA block plus another group's turn covers the gather latency
    f32x4 c00[2][A_IT], c01[2][A_IT], c10[2][A_IT], c11[2][A_IT];
    float w00[2][A_IT], w01[2][A_IT], w10[2][A_IT], w11[2][A_IT];
    f32x4 rb[2][B_IT];                                // fp32 MFMA path: staged weights
    typedef __bf16 dcn_bf16x8 __attribute__((ext_vector_type(8)));
    dcn_bf16x8 bfr[2][NP][2][TN];                     // bf16 products: B fragments [set][plane][k-step][column tile]

    // ---- unit decode through a table in LDS.  A K-chunk is two units (uu = 0 / 1 by lane); unit u = ((g * KK) + tap) * cgq + cq.
    // Everything a lane needs from (g, tap, cq) -- offset / mask / flow word offsets, the tap's (ky, kx) * dilation, the
    // channel base -- is tabulated once per workgroup; the K loop then does one 32-byte LDS read per item instead of
    // three runtime integer divisions (the kernel was VALU-bound on them: -19 % without).
    for (int u = threadIdx.x; u < p.units; u += NG * KS) {
        const int g = u / (p.KK * p.cgq);
        const int rem = u - g * (p.KK * p.cgq);
        const int tap = rem / p.cgq, cq = rem - tap * p.cgq;
        const int ky = tap / p.KW, kx = tap - ky * p.KW;
        int* e = utab + u * 8;
        e[0] = (g * 2 * p.KK + 2 * tap) * 4;
        e[1] = (g * p.KK + tap) * 4;
        e[2] = (g * 2 >= p.dg) ? 8 : 0;
        e[3] = ky * p.dil;
        e[4] = kx * p.dil;
        e[5] = (g * p.cg + cq * 16) * 4;
        e[6] = 0;
        e[7] = 0;
    }
    __syncthreads();
    const unsigned c_W = (unsigned)__builtin_amdgcn_readfirstlane(p.W);
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    // ---- sampler threads: slot = (pixel row, unit in chunk).  SLOTS / 64 whole waves of the group fetch the raw offset /
    // mask / flow words of a chunk AND turn them into the sample's four corner addresses and bilinear weights, once per
    // (row, unit): the 16 channels of a unit are 2 (bf16 sources) or 4 (fp32) 16-byte items that used to repeat this
    // arithmetic (~100 VALU instructions each; the kernel is VALU-bound: 4 MFMAs against ~600 VALU cycles per wave and
    // chunk in the bf16 instantiation).  The sampler waves rotate with the K group so that every SIMD hosts one.
    // Measured (tools/dcn_bench.py, dcn_bench_x.py): fp32 60x108 tile 5 64.6 -> 60.6 us; bf16 180x324 tile 6 unchanged
    // (195 -> 199 us): there the corner fetches themselves (16-byte pieces of 512-byte pixels, L1 working set > 32 KB) bound it.
    float* const graw = sraw + kg * (2 * SLOTS * 8);
    constexpr int NWG = WGM * WGN, LW = SLOTS / 64;
    const int lw = (KS == 1) ? wave : (wave - (kg * LW) % NWG + NWG) % NWG;
    const bool loader = lw < LW;                           // wave-uniform
    const int ltid = lw * 64 + lane;
    unsigned l_po = 0, l_pm = 0, l_pf = 0;
    int l_by = 0, l_bx = 0, l_imgrow = 0;
    bool l_ok = false;
    {
        int img, oy, ox;
        if (p.sw) {                                        // block layout: tile_m = (img, block y, block x), row = (y, x) inside 8 wide
            const int r = ltid >> 1;
            img = tile_m / (p.swX * p.swY);
            const int rem = tile_m - img * (p.swX * p.swY);
            const int ty = rem / p.swX, tx = rem - ty * p.swX;
            oy = ty * (BM / 8) + (r >> 3);
            ox = tx * 8 + (r & 7);
            l_ok = loader && oy < p.Ho && ox < p.Wo;
            if (!l_ok) { img = 0; oy = 0; ox = 0; }
        } else {
            const int m = m0 + (ltid >> 1);
            l_ok = loader && m < p.M;
            const int mm = l_ok ? m : 0;
            img = mm / HoWo;
            const int rem = mm - img * HoWo;
            oy = rem / p.Wo;
            ox = rem - oy * p.Wo;
        }
        const unsigned pix = (unsigned)((img * p.Ho + oy) * p.Wo + ox);
        l_po = pix * (unsigned)p.off_ld * 4u;
        l_pm = pix * (unsigned)p.msk_ld * 4u;
        l_pf = pix * 16u;
        l_imgrow = img * p.H;
        l_by = oy * p.stride - p.pad;
        l_bx = ox * p.stride - p.pad;
    }
    constexpr unsigned OOBS = 0x80000000u;                 // out-of-range corner: stays out of range after + part * 16 (sources < 2 GiB)
    f32x2 l_d = {0.f, 0.f}, l_f = {0.f, 0.f};
    float l_mk = 0.f;
    auto load_offsets = [&](int kt) {
        if (loader) {
            const int u = 2 * kt + (ltid & 1);
            const bool ok = kt < KT && u < p.units && l_ok;
            const i32x4 e = *reinterpret_cast<const i32x4*>(utab + (ok ? u : 0) * 8);
            l_d = buf_load2(r_off, ok ? l_po + (unsigned)e[0] : OOB);
            l_mk = buf_load1(r_msk, ok ? l_pm + (unsigned)e[1] : OOB);
            l_f = buf_load2(r_flw, ok ? l_pf + (unsigned)e[2] : OOB);
        }
    };
    // the sampler's half: raw words of chunk kt (in its registers) -> slot {a00, a01, a10, a11, w00, w01, w10, w11}
    auto store_offsets = [&](int buf, int kt) {
        if (loader) {
            const int u = 2 * kt + (ltid & 1);
            const bool uok = kt < KT && u < p.units && l_ok;
            const int* e = utab + (uok ? u : 0) * 8;
            const int dyk = e[3], dxk = e[4];
            // both units of a chunk read the same source (host guarantees an even unit count in source 0)
            const bool second_src = (__builtin_amdgcn_readfirstlane(kt) * 2) >= p.units0;
            const unsigned cbase4 = second_src ? (unsigned)p.c[0] * (unsigned)SB : 0u;
            unsigned ld4 = (unsigned)(second_src ? p.ld[1] : p.ld[0]) * (unsigned)SB;
            unsigned ch = (unsigned)(S16 ? e[5] >> 1 : e[5]) - cbase4;
            if (S16 && p.planar) {                         // [group][pixel][16 bf16]: pixel stride 32 bytes, a plane per group
                ld4 = 32u;
                ch = (ch >> 5) * p.plane_bytes;
            }
            float dy = l_d[0], dx = l_d[1], mk = l_mk;
            if (p.flows) {
                // tanh / sigmoid through v_exp_f32 + v_rcp_f32 (abs error ~2e-7: <1e-5 px on the residual offset)
                dy = p.max_residue * fast_tanh(dy) + l_f[1];       // flip: dy takes the v (y) component
                dx = p.max_residue * fast_tanh(dx) + l_f[0];
                mk = fast_sigmoid(mk);
            }
            const float py = (float)(l_by + dyk) + dy;
            const float px = (float)(l_bx + dxk) + dx;
            const bool inside = uok && py > -1.f && px > -1.f && py < (float)p.H && px < (float)p.W;
            const float fy = floorf(py), fx = floorf(px);
            const int y0 = (int)fy, x0 = (int)fx, y1 = y0 + 1, x1 = x0 + 1;
            const float ly = py - fy, lx = px - fx, hy = 1.f - ly, hx = 1.f - lx;
            const bool vy0 = inside && y0 >= 0, vy1 = inside && y1 <= p.H - 1, vx0 = x0 >= 0, vx1 = x1 <= p.W - 1;
            const float mm = inside ? mk : 0.f;
            // 24-bit multiplies (full rate): pixel indices and pixel strides are < 2^24 (checked on the host).  The products
            // are formed from the row y1 and column x1, which are >= 0 whenever the sample is inside; the y0 / x0 addresses
            // follow by subtraction (they may wrap when y0 or x0 is -1 -- those corners are replaced by the sentinel)
            const unsigned r1 = __umul24((unsigned)(l_imgrow + y1), c_W);
            const unsigned r0 = r1 - c_W;
            const unsigned a01 = __umul24(r0 + (unsigned)x1, ld4) + ch;
            const unsigned a11 = __umul24(r1 + (unsigned)x1, ld4) + ch;
            const unsigned a00 = a01 - ld4, a10 = a11 - ld4;
            const u32x4 ad = {(vy0 && vx0) ? a00 : OOBS, (vy0 && vx1) ? a01 : OOBS, (vy1 && vx0) ? a10 : OOBS, (vy1 && vx1) ? a11 : OOBS};
            const f32x4 ww = {hy * hx * mm, hy * lx * mm, ly * hx * mm, ly * lx * mm};
            float* d = graw + (buf * SLOTS + ltid) * 8;
            *reinterpret_cast<u32x4*>(d) = ad;
            *reinterpret_cast<f32x4*>(d + 4) = ww;
        }
    };
    // the items' half: a slot's addresses + this item's 16-byte part -> corner fetches of chunk kt
    auto issue_corners = [&](int kt, int S, int rbuf) {
        // (kt is wave-uniform; said so explicitly: hipcc otherwise selected the resource with v_cndmask in one of the two unrolled
        //  halves of the K loop and wrapped each of the four corner loads in a waterfall loop -- ~45 instructions per chunk)
        const bool second_src = (__builtin_amdgcn_readfirstlane(kt) * 2) >= p.units0;
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(second_src ? p.src[1] : p.src[0], second_src ? p.src_bytes[1] : p.src_bytes[0]);
#pragma unroll
        for (int ia = 0; ia < A_IT; ++ia) {
            const float* slot = graw + (rbuf * SLOTS + it_row[ia] * 2 + it_uu[ia]) * 8;
            const u32x4 ad = *reinterpret_cast<const u32x4*>(slot);
            const f32x4 ww = *reinterpret_cast<const f32x4*>(slot + 4);
            const unsigned part = (unsigned)it_c4[ia] * 16u;
            w00[S][ia] = ww[0]; w01[S][ia] = ww[1]; w10[S][ia] = ww[2]; w11[S][ia] = ww[3];
            c00[S][ia] = buf_load4(rs, it_ok[ia] ? ad[0] + part : OOBS);
            c01[S][ia] = buf_load4(rs, it_ok[ia] ? ad[1] + part : OOBS);
            c10[S][ia] = buf_load4(rs, it_ok[ia] ? ad[2] + part : OOBS);
            c11[S][ia] = buf_load4(rs, it_ok[ia] ? ad[3] + part : OOBS);
        }
    };
    auto load_w = [&](int kt, int S) {
        if constexpr (BF) {
            // straight into the MFMA's B operand registers (chunks past the end: out of range = zeros)
            const unsigned vo = kt < KT ? bf_lane + (unsigned)kt * b_step : OOB;
#pragma unroll
            for (int pl = 0; pl < NP; ++pl)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        bfr[S][pl][kk][tn] = __builtin_bit_cast(dcn_bf16x8, __builtin_amdgcn_raw_buffer_load_b128(
                            r_w, (int)vo, (int)((unsigned)(pl * 2) * bf_oct2 + (unsigned)kk * bf_oct2 + bf_tn[tn]), 0));
        } else {
            const bool tile_ok = kt < KT;
#pragma unroll
            for (int ib = 0; ib < B_IT; ++ib)
                rb[S][ib] = buf_load4(r_w, (b_off[ib] == OOB || !tile_ok) ? OOB : b_off[ib] + (unsigned)kt * b_step);
        }
    };
    auto store_tile = [&](int buf, int S) {
        float* sA = sbase + buf * STAGE;
        float* sB = sA + BM * LDA;                     // (fp32 MFMA path only)
#pragma unroll
        for (int ia = 0; ia < A_IT; ++ia) {
            if (A_ITEMS % NG == 0 || (tid + ia * NG) < A_ITEMS) {
                if constexpr (S16) {
                    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
                    const u32x4 q00 = __builtin_bit_cast(u32x4, c00[S][ia]), q01 = __builtin_bit_cast(u32x4, c01[S][ia]),
                                q10 = __builtin_bit_cast(u32x4, c10[S][ia]), q11 = __builtin_bit_cast(u32x4, c11[S][ia]);
                    bf16x8 hv;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float lo = __builtin_bit_cast(float, q00[e] << 16) * w00[S][ia] + __builtin_bit_cast(float, q01[e] << 16) * w01[S][ia] +
                                         __builtin_bit_cast(float, q10[e] << 16) * w10[S][ia] + __builtin_bit_cast(float, q11[e] << 16) * w11[S][ia];
                        const float hi = __builtin_bit_cast(float, q00[e] & 0xFFFF0000u) * w00[S][ia] + __builtin_bit_cast(float, q01[e] & 0xFFFF0000u) * w01[S][ia] +
                                         __builtin_bit_cast(float, q10[e] & 0xFFFF0000u) * w10[S][ia] + __builtin_bit_cast(float, q11[e] & 0xFFFF0000u) * w11[S][ia];
                        hv[2 * e] = (__bf16)lo;
                        hv[2 * e + 1] = (__bf16)hi;
                    }
                    *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(sA) + it_row[ia] * LDA16 + it_uu[ia] * 16 + it_c4[ia] * 8) = hv;
                    continue;
                }
                const f32x4 v = c00[S][ia] * w00[S][ia] + c01[S][ia] * w01[S][ia] + c10[S][ia] * w10[S][ia] + c11[S][ia] * w11[S][ia];
                if constexpr (X3) {
                    // hi / mid / lo of the four blended values (common.h, e2_split2): 8-byte writes into the three A planes
                    float xv[4];                         // (scalar copies: element access through a vector reference)
#pragma unroll
                    for (int e = 0; e < 4; ++e) xv[e] = v[e];
                    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                    u32x2 ph, pm, pl;
                    {
                        unsigned h0, m0, l0, h1, m1, l1;
                        e2_split2(xv[0], xv[1], h0, m0, l0);
                        e2_split2(xv[2], xv[3], h1, m1, l1);
                        ph[0] = h0; ph[1] = h1; pm[0] = m0; pm[1] = m1; pl[0] = l0; pl[1] = l1;
                    }
                    __bf16* a16 = reinterpret_cast<__bf16*>(sA) + it_row[ia] * LDA16 + it_uu[ia] * 16 + it_c4[ia] * 4;
                    *reinterpret_cast<u32x2*>(a16) = ph;
                    *reinterpret_cast<u32x2*>(a16 + BM * LDA16) = pm;
                    *reinterpret_cast<u32x2*>(a16 + 2 * BM * LDA16) = pl;
                } else if (BF) {
                    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
                    bf16x4 hv = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    *reinterpret_cast<bf16x4*>(reinterpret_cast<__bf16*>(sA) + it_row[ia] * LDA16 + it_uu[ia] * 16 + it_c4[ia] * 4) = hv;
                } else {
                    *reinterpret_cast<f32x4*>(sA + it_row[ia] * LDA + it_uu[ia] * 16 + it_c4[ia] * 4) = v;
                }
            }
        }
        if constexpr (!BF) {
#pragma unroll
            for (int ib = 0; ib < B_IT; ++ib) {
                const int f = tid + ib * NG;
                if (B_F4 % NG == 0 || f < B_F4) *reinterpret_cast<f32x4*>(sB + f * 4) = rb[S][ib];
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

    // this group's chunks: kt = kg + it*KS.  Chunks past the end read zeros, so no guards are needed.
    // chunks: kt = kg + it*KS.  Chunks past the end read zeros, so no guards are needed and the iteration count is rounded
    // up to even (the loop is unrolled by two so that the register sets alternate with compile-time indices).
    const int nIter = ((KT + KS - 1) / KS + 1) & ~1;
    // software pipeline over this group's chunk sequence j = 0, 1, 2 ... (chunk kg + j KS); at the top of step i:
    //   tile i in the LDS ring, corners + weights of chunk i+1 in flight in register set (i+1)&1,
    //   raw words of chunk i+2 in sraw buffer i&1, raw words of chunk i+3 in the loader's registers
    load_offsets(kg);
    store_offsets(0, kg);
    load_offsets(kg + KS);
    __syncthreads();
    issue_corners(kg, 0, 0);
    load_w(kg, 0);
    store_offsets(1, kg + KS);
    load_offsets(kg + 2 * KS);
    store_tile(0, 0);
    __syncthreads();
    issue_corners(kg + KS, 1, 1);
    load_w(kg + KS, 1);
    store_offsets(0, kg + 2 * KS);
    load_offsets(kg + 3 * KS);
    __syncthreads();

    int cur = 0;
    for (int it = 0; it < nIter; it += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int kt = kg + (it + par) * KS;
            issue_corners(kt + 2 * KS, par, par);            // raw words of chunk i+2 from sraw buffer i&1
            if constexpr (!BF) load_w(kt + 2 * KS, par);     // (bf16 products: behind the MFMAs, into the fragment set they have just read)
            if constexpr (X3) {
                typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
                const __bf16* sA16 = reinterpret_cast<const __bf16*>(sbase + cur * STAGE);
                const int li = lane & 31, lh = lane >> 5;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8 a[3][TM], b[3][TN];
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
                            a[pl][tm] = *reinterpret_cast<const bf16x8*>(sA16 + pl * (BM * LDA16) + ((wm * TM + tm) * 32 + li) * LDA16 + (2 * kk + lh) * 8);
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) b[pl][tn] = bfr[par][pl][kk][tn];
                    }
                    // smallest terms first: (lo, hi) (hi, lo) (mid, mid) (mid, hi) (hi, mid) (hi, hi)
                    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
                    for (int t6 = 0; t6 < 6; ++t6)
#pragma unroll
                        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                            for (int tn = 0; tn < TN; ++tn)
                                acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PA[t6]][tm], b[PB[t6]][tn], acc[tm][tn], 0, 0, 0);
                }
            } else if (BF) {
                typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
                const __bf16* sA16 = reinterpret_cast<const __bf16*>(sbase + cur * STAGE);
                const int li = lane & 31, lh = lane >> 5;
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    bf16x8 a[TM], b[TN];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
                        a[tm] = *reinterpret_cast<const bf16x8*>(sA16 + ((wm * TM + tm) * 32 + li) * LDA16 + (2 * kk + lh) * 8);
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) b[tn] = bfr[par][0][kk][tn];
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
                }
            } else {
                mma_ktile<TM, TN, BK, LDA, BN>(sbase + cur * STAGE, sbase + cur * STAGE + BM * LDA, acc, wm * TM * 32,
                                               wn * TN * 32, lane);
            }
            if constexpr (BF) load_w(kt + 2 * KS, par);      // chunk i+2's B fragments: a whole trip to land
            store_tile(cur ^ 1, par ^ 1);                    // chunk i+1
            store_offsets(par ^ 1, kt + 3 * KS);             // chunk i+3 (that buffer's readers passed the last barrier)
            load_offsets(kt + 4 * KS);
            __syncthreads();
            cur ^= 1;
        }
    }

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
                int m = m0 + (wm * TM + tm) * 32 + row;
                bool ok = m < p.M;
                if (p.sw) {
                    const int rr = (wm * TM + tm) * 32 + row;
                    const int img = tile_m / (p.swX * p.swY);
                    const int rem = tile_m - img * (p.swX * p.swY);
                    const int ty = rem / p.swX, tx = rem - ty * p.swX;
                    const int oy = ty * (BM / 8) + (rr >> 3), ox = tx * 8 + (rr & 7);
                    ok = oy < p.Ho && ox < p.Wo;
                    m = (img * p.Ho + oy) * p.Wo + ox;
                }
                if (ok) {
                    const long long o = (long long)m * p.dst_ld + p.dst_coff + n;
                    if (p.dst_bf16) reinterpret_cast<__bf16*>(p.dst)[o] = (__bf16)(acc[tm][tn][r] + bv);
                    else p.dst[o] = acc[tm][tn][r] + bv;
                }
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

// bf16 weights of the BF variant: [chunk][4 k-octets][Npad][8], k inside a chunk = (unit in chunk) * 16 + channel, like above
__global__ void pack_dcn_weight_bf16_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int Cout, int C, int KK,
                                            int cg, int Npad, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = (int)(idx & 7);
    long long rem = idx >> 3;
    const int n = (int)(rem % Npad);
    const int ko = (int)(rem / Npad);
    const int k = ko * 8 + e;
    const int u = k >> 4, c = k & 15;
    const int cgq = cg / 16;
    const int g = u / (KK * cgq);
    const int r2 = u - g * (KK * cgq);
    const int tap = r2 / cgq, cq = r2 - tap * cgq;
    const int ch = g * cg + cq * 16 + c;
    float v = 0.f;
    if (ch < C && n < Cout && g * cg < C) v = w[((long long)n * C + ch) * KK + tap];
    wp[idx] = (__bf16)v;
}

// split-operand weights (X3): [chunk][plane hi, mid, lo][4 k-octets][Npad][8] bf16; `total` counts the fp32 values
__global__ void pack_dcn_weight_x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cout, int C, int KK,
                                          int cg, int Npad, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = (int)(idx & 7);
    long long rem = idx >> 3;
    const int n = (int)(rem % Npad);
    const int ko = (int)(rem / Npad);
    const int k = ko * 8 + e;
    const int u = k >> 4, c = k & 15;
    const int cgq = cg / 16;
    const int g = u / (KK * cgq);
    const int r2u = u - g * (KK * cgq);
    const int tap = r2u / cgq, cq = r2u - tap * cgq;
    const int ch = g * cg + cq * 16 + c;
    float v = 0.f;
    if (ch < C && n < Cout && g * cg < C) v = w[((long long)n * C + ch) * KK + tap];
    const unsigned xb = __builtin_bit_cast(unsigned, v);
    const float r = v - __builtin_bit_cast(float, xb & 0xFFFF0000u);
    const unsigned rb = __builtin_bit_cast(unsigned, r);
    const float r2 = r - __builtin_bit_cast(float, rb & 0xFFFF0000u);
    const int chunk = ko >> 2, oct = ko & 3;
    const long long plane = 4LL * Npad * 8;
    unsigned short* o = wp + (long long)chunk * 3 * plane + ((long long)oct * Npad + n) * 8 + e;
    o[0] = (unsigned short)(xb >> 16);
    o[plane] = (unsigned short)(rb >> 16);
    o[2 * plane] = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
}

long long dcn_packed_size(int Cout, int C, int KH, int KW) {
    const int units = (C / 16) * KH * KW;
    const int KT = (units + 1) / 2;
    return (long long)KT * 32 * round_up(Cout, 32);
}

template <int BM, int BN, int WGM, int WGN, int KS>
int launch_dcn(DcnParams& p, hipStream_t st, bool bf, bool s16, bool x3 = false) {
    p.tilesM = cdiv(p.M, BM);
    if (p.sw) {
        p.swX = cdiv(p.Wo, 8);
        p.swY = cdiv(p.Ho, BM / 8);
        p.tilesM = p.N * p.swX * p.swY;
    }
    p.tilesN = cdiv(p.Cout, BN);
    if (x3) {
        // three bf16 planes per operand: two ring stages per K group must fit the LDS beside the offset slots
        if constexpr (mdcn_fits_lds<BM, BN, WGM, WGN, KS, true, true>())
            hipLaunchKernelGGL((mdcn_kernel<BM, BN, WGM, WGN, KS, true, false, true>), dim3(p.tilesM * p.tilesN), dim3(64 * WGM * WGN * KS), 0, st, p);
        else {
            e2fgvi_set_error("mdcn: this tile does not fit the LDS with split operands");
            return E2FGVI_EUNSUP;
        }
    } else if (s16 || bf) {
        if constexpr (mdcn_fits_lds<BM, BN, WGM, WGN, KS, true, false>()) {
            if (s16)
                hipLaunchKernelGGL((mdcn_kernel<BM, BN, WGM, WGN, KS, true, true>), dim3(p.tilesM * p.tilesN), dim3(64 * WGM * WGN * KS), 0, st, p);
            else
                hipLaunchKernelGGL((mdcn_kernel<BM, BN, WGM, WGN, KS, true, false>), dim3(p.tilesM * p.tilesN), dim3(64 * WGM * WGN * KS), 0, st, p);
        } else {
            e2fgvi_set_error("mdcn: this tile does not fit the LDS with bf16 operands");
            return E2FGVI_EUNSUP;
        }
    } else {
        // fp32 MFMA: the weights are staged through LDS
        if constexpr (mdcn_fits_lds<BM, BN, WGM, WGN, KS, false, false>())
            hipLaunchKernelGGL((mdcn_kernel<BM, BN, WGM, WGN, KS, false, false>), dim3(p.tilesM * p.tilesN), dim3(64 * WGM * WGN * KS), 0, st, p);
        else {
            e2fgvi_set_error("mdcn: this tile does not fit the LDS with fp32 operands");
            return E2FGVI_EUNSUP;
        }
    }
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

/* bf16 weights for mfma_dtype = E2FGVI_BF16: same element count as the fp32 packing, 2 bytes each */
extern "C" int e2fgvi_pack_dcn_weight_bf16(const float* w, void* wpacked, int32_t Cout, int32_t C, int32_t KH, int32_t KW,
                                           int32_t deform_groups, void* stream) {
    E2_REQUIRE(w && wpacked, E2FGVI_EINVAL, "pack_dcn_weight_bf16: null pointer");
    E2_REQUIRE(Cout > 0 && C > 0 && deform_groups > 0 && C % deform_groups == 0 && (C / deform_groups) % 16 == 0,
               E2FGVI_EUNSUP, "pack_dcn_weight_bf16: channels per deform group must be a multiple of 16");
    const long long total = dcn_packed_size(Cout, C, KH, KW);
    hipLaunchKernelGGL(pack_dcn_weight_bf16_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (__bf16*)wpacked, Cout, C, KH * KW, C / deform_groups, round_up(Cout, 32), total);
    E2_LAUNCH_CHECK("pack_dcn_weight_bf16");
    return 0;
}

/* split-operand weights for mfma_dtype = E2FGVI_BF16X3: 3 x e2fgvi_packed_dcn_weight_size elements of 2 bytes (ABI 7) */
extern "C" int e2fgvi_pack_dcn_weight_x3(const float* w, void* wpacked, int32_t Cout, int32_t C, int32_t KH, int32_t KW,
                                         int32_t deform_groups, void* stream) {
    E2_REQUIRE(w && wpacked, E2FGVI_EINVAL, "pack_dcn_weight_x3: null pointer");
    E2_REQUIRE(Cout > 0 && C > 0 && deform_groups > 0 && C % deform_groups == 0 && (C / deform_groups) % 16 == 0,
               E2FGVI_EUNSUP, "pack_dcn_weight_x3: channels per deform group must be a multiple of 16");
    const long long total = dcn_packed_size(Cout, C, KH, KW);
    hipLaunchKernelGGL(pack_dcn_weight_x3_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (unsigned short*)wpacked, Cout, C, KH * KW, C / deform_groups, round_up(Cout, 32), total);
    E2_LAUNCH_CHECK("pack_dcn_weight_x3");
    return 0;
}

extern "C" int e2fgvi_mdcn_nhwc(const e2fgvi_mdcn_desc* d, void* stream) {
    E2_REQUIRE(d, E2FGVI_EINVAL, "mdcn: null descriptor");
    E2_REQUIRE(d->nsrc == 1 || d->nsrc == 2, E2FGVI_EINVAL, "mdcn: nsrc must be 1 or 2");
    DcnParams p;
    int C = 0;
    E2_REQUIRE(d->src_dtype == E2FGVI_F32 || (d->src_dtype == E2FGVI_BF16 && d->mfma_dtype == E2FGVI_BF16), E2FGVI_EINVAL,
               "mdcn: src_dtype must be E2FGVI_F32, or E2FGVI_BF16 together with mfma_dtype = E2FGVI_BF16");
    E2_REQUIRE(d->mfma_dtype == E2FGVI_F32 || d->mfma_dtype == E2FGVI_BF16 || d->mfma_dtype == E2FGVI_BF16X3, E2FGVI_EINVAL,
               "mdcn: mfma_dtype must be E2FGVI_F32, E2FGVI_BF16 or E2FGVI_BF16X3");
    const bool x3 = d->mfma_dtype == E2FGVI_BF16X3;
    const bool s16 = d->src_dtype == E2FGVI_BF16;
    const int sb = s16 ? 2 : 4;
    for (int s = 0; s < 2; ++s) { p.src[s] = nullptr; p.ld[s] = 0; p.c[s] = 0; }
    for (int s = 0; s < d->nsrc; ++s) {
        E2_REQUIRE(d->src[s] && d->src_c[s] > 0 && d->src_ld[s] >= d->src_c[s] && d->src_ld[s] % (16 / sb) == 0 &&
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
    E2_REQUIRE(d->off_ld % 2 == 0, E2FGVI_EINVAL, "mdcn: off_ld must be even (dy, dx are fetched as one 8-byte word)");
    E2_REQUIRE((C / 16) * d->KH * d->KW <= MAX_UNITS, E2FGVI_EUNSUP, "mdcn: C/16 * KH * KW exceeds the %d-entry unit table", MAX_UNITS);
    E2_REQUIRE((long long)d->N * d->H * d->W + d->W < (1 << 24) && d->src_ld[0] * 4 < (1 << 24) &&
                   (d->nsrc == 1 || d->src_ld[1] * 4 < (1 << 24)),
               E2FGVI_EUNSUP, "mdcn: more than 2^24 input pixels per call (split the batch)");
    E2_REQUIRE(!d->flows || d->deform_groups % 2 == 0, E2FGVI_EINVAL, "mdcn: fused flows need an even group count");
    E2_REQUIRE(!d->src_planar || (s16 && cg == 16), E2FGVI_EUNSUP, "mdcn: src_planar needs bf16 sources and 16 channels per deform group");
    p.planar = d->src_planar ? 1 : 0;
    p.plane_bytes = (unsigned)((long long)d->N * d->H * d->W * 32);
    if (p.planar) { p.ld[0] = d->src_c[0]; p.ld[1] = d->nsrc == 2 ? d->src_c[1] : p.ld[0]; }      // byte bounds below: c / 16 planes
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
    p.dst = d->dst; p.dst_ld = d->dst_ld; p.dst_coff = d->dst_coff; p.dst_bf16 = d->dst_dtype == E2FGVI_BF16;
    // buffer bounds (all < 4 GiB) and the source split
    {
        const long long P = (long long)d->N * d->Ho * d->Wo;
        const long long sb0 = (long long)d->N * d->H * d->W * p.ld[0] * sb, sb1 = (long long)d->N * d->H * d->W * p.ld[1] * sb;
        const long long ob = P * d->off_ld * 4, mb = P * d->mask_ld * 4,
                        wb = dcn_packed_size(d->Cout, C, d->KH, d->KW) * (d->mfma_dtype == E2FGVI_BF16 ? 2 : x3 ? 6 : 4);
        E2_REQUIRE(sb0 < 2147483392LL && sb1 < 2147483392LL && ob < 4294967295LL && mb < 4294967295LL && wb < 4294967295LL,
                   E2FGVI_EUNSUP, "mdcn: a source spans >= 2 GiB (or offsets / masks >= 4 GiB); buffer addressing needs less (split the batch)");
        p.src_bytes[0] = (unsigned)sb0; p.src_bytes[1] = (unsigned)sb1;
        p.off_bytes = (unsigned)ob; p.msk_bytes = (unsigned)mb; p.flw_bytes = (unsigned)(P * 16); p.w_bytes = (unsigned)wb;
        // the mask may alias the offset tensor (raw conv_offset layout): bound it by what is left behind its base
        if (d->mask >= d->offset && (const char*)d->mask < (const char*)d->offset + ob)
            p.msk_bytes = (unsigned)(ob - ((const char*)d->mask - (const char*)d->offset));
        p.units0 = (d->nsrc == 2) ? (d->src_c[0] / 16) * p.KK : p.units;
        E2_REQUIRE(d->nsrc == 1 || p.units0 % 2 == 0, E2FGVI_EUNSUP, "mdcn: odd number of 16-channel units in source 0");
    }
    int tile = d->tile;
    p.sw = 0; p.swX = p.swY = 1;
    if (tile >= 100) { p.sw = 1; tile -= 100; }          // 101 ... 106: the same tiles on 8 x (BM / 8) pixel blocks
    if (!tile) {
        const long long b64 = (long long)cdiv(p.M, 64) * cdiv(p.Cout, 128);
        tile = b64 >= 512 ? 1 : (b64 >= 256 ? 2 : 5);
        // bf16 product: the 64-row tile with two K groups (tools/dcn_bench_x.py at 180x324: 195 vs 203 us with bf16 sources,
        // 221 vs 225 with fp32 sources)
        // Its 64 rows as an 8 x 8 pixel block (tile 106: the corner fetches of a tile fall into a (8 + 2r)^2 neighbourhood instead
        // of (64 + 2r) x (1 + 2r) pixels) is 9 % faster on i.i.d. random 3-pixel offsets (177.9 vs 196.2 us) and exactly neutral
        // inside the forward, where the offsets follow the smooth flow field (34.46 / 34.57 vs 34.42 / 34.58 ms, same box,
        // profiles/r02_dcn_sampler.txt): off by default, tile codes 101 ... 106 select it
        // (planar sources: the single K group wins, 143 vs 156 us on a smooth offset field, 170 vs 178 on random offsets)
        if (tile == 1 && d->mfma_dtype == E2FGVI_BF16 && !p.planar) {
            tile = 6;
        }
    }
    const bool bf = d->mfma_dtype == E2FGVI_BF16;
    // (round 5: with the weights out of the LDS the three-K-group tile fits with split operands too: 53.0 vs 56.0 us at 60x108,
    //  profiles/r05_dcn_tiles.txt)
    if (tile == 1) return launch_dcn<64, 128, 2, 2, 1>(p, (hipStream_t)stream, bf, s16, x3);
    if (tile == 2) return launch_dcn<32, 128, 1, 4, 1>(p, (hipStream_t)stream, bf, s16, x3);
    if (tile == 3) return launch_dcn<32, 64, 1, 2, 1>(p, (hipStream_t)stream, bf, s16, x3);
    if (tile == 4) return launch_dcn<32, 128, 1, 4, 2>(p, (hipStream_t)stream, bf, s16, x3);
    if (tile == 5) return launch_dcn<32, 128, 1, 4, 3>(p, (hipStream_t)stream, bf, s16, x3);
    if (tile == 6) return launch_dcn<64, 128, 2, 2, 2>(p, (hipStream_t)stream, bf, s16, x3);
    // four K groups (round 5: with the weights out of the LDS a stage is the A slab alone): bf16 products only
    if (tile == 7 && (bf || x3)) return launch_dcn<32, 128, 1, 4, 4>(p, (hipStream_t)stream, bf, s16, x3);
    e2fgvi_set_error("mdcn: unknown tile %d", tile);
    return E2FGVI_EINVAL;
}
