// Winograd F(FY x 4, 3x3) fp32 convolution on the fp32 MFMA pipe (gfx950), FY in {2, 4}: the wide-tile sibling of
// conv_wino.hip's F(2x2,3x3).  Same operator (3x3, stride 1, pad 1, NHWC, virtual concat of up to 4 sources, groups),
// same structure, fewer multiplies per output:
//
//   F(2x2,3x3): 16 positions per 2x2 outputs = 4    multiplies per output and input channel   (direct: 9)
//   F(2x4,3x3): 24 positions per 2x4 outputs = 3                                               (this file, FY = 2)
//   F(4x4,3x3): 36 positions per 4x4 outputs = 2.25                                            (this file, FY = 4)
//
//   Y = A_y^T [ (G_y g G_x^T) (.) (B_y^T d B_x) ] A_x     per (FY+2) x 6 input patch d, 3x3 filter g; the x side is always
//   F(4,3) (Lavin & Gray's matrices, interpolation points 0, +-1, +-2, inf), the y side F(2,3) or F(4,3)
//   = (FY+2)*6 independent GEMMs  M_a[tile, cout] = sum_cin V_a[tile, cin] * U_a[cin, cout].
//
// Workgroup = 2 (FY+2) waves = one block of 8 x 4 Winograd tiles (8 FY x 16 pixels) x BN output channels.  Wave w owns the
// transform row xi = w >> 1 and three of its six column positions (nu = 3 (w & 1) + 0..2): 3 x (BN/32) 32x32
// accumulators.  Per 8-channel chunk of the input:
//   * the raw (8 FY + 2) x 18-pixel halo patch is staged ONCE in LDS (global -> registers -> LDS one stage = two chunks
//     ahead); pixel columns live in four planes by (column mod 4), so that the stride-4 patch reads of the 32 tiles of an
//     MFMA operand are bank-conflict free (row pitch chosen so that FY * pitch = 4 mod 16 sixteen-byte units);
//   * every lane builds its MFMA A operands straight from the patch: the rows with a non-zero B_y^T coefficient of its xi
//     x 5 of the 6 patch columns (ds_read_b128, 4 channels each), row combination, then the three column combinations;
//   * B operands (pre-transformed weights of the wave's three positions) go global -> registers as exactly the lane's
//     operand quad, issued one chunk ahead through pinned loads + explicit s_waitcnt (as conv_wino.hip).
// Epilogue: the positions meet in LDS (16 tiles at a time), every thread applies A_y^T M A_x for one (tile, output row,
// 4 couts) item, adds the bias and the residual, applies the activation (or the DCN offset / mask post-processing) and
// stores 4 pixels x 16 bytes.
//
// Packed weights: [group][chunk][a = (FY+2)*6][kq = 2][Npad][4]  (chunk = 8 input channels in concat order, sources
// padded to 8; Npad = Cout_g rounded up to 32), produced by e2fgvi_pack_winograd4_weight.
//
// Numerics (fp32, random data, 512 input channels; relative to the output rms): direct 7.7e-6, F(2x2) 2.9e-6,
// F(2x4) 6.5e-6, F(4x4) 1.9e-5.
#include "common.h"
#include <utility>

// register claims behind the K loop (conv_wino.hip)
#ifndef E2_CLAIM_AFTER_LOOP
#define E2_CLAIM_AFTER_LOOP(r) asm volatile("" : "+v"(r))
#endif

namespace {

struct W4Params {
    const float* src[E2FGVI_MAX_SRC];
    int ld[E2FGVI_MAX_SRC];
    int coff[E2FGVI_MAX_SRC];
    int cpg[E2FGVI_MAX_SRC];
    unsigned src_bytes[E2FGVI_MAX_SRC];
    int nsrc;
    int N, H, W;
    int Cout, Cout_g, Npad;
    int blocksY, blocksX, tilesN, nblk;
    int nchunks;
    unsigned wgroup_bytes;
    long long wgroup_elems;
    const float* w;
    const float* bias;
    const float* res;       // residual [pixel][res_ld] (+ res_coff), or the per-pixel flows [pixel][4] of ACT_DCNPOST
    int res_ld, res_coff;
    float* dst;
    int dst_ld, dst_coff;
    int act;
    float slope;
    int vec_store;
};

// ---- transform matrices (compile-time after unrolling; zero coefficients are skipped, not multiplied)
__host__ __device__ constexpr float bt4(int r, int c) {          // B^T of F(4,3): 6 x 6
    return r == 0   ? (c == 0 ? 4.f : c == 2 ? -5.f : c == 4 ? 1.f : 0.f)
           : r == 1 ? (c == 1 ? -4.f : c == 2 ? -4.f : c == 3 ? 1.f : c == 4 ? 1.f : 0.f)
           : r == 2 ? (c == 1 ? 4.f : c == 2 ? -4.f : c == 3 ? -1.f : c == 4 ? 1.f : 0.f)
           : r == 3 ? (c == 1 ? -2.f : c == 2 ? -1.f : c == 3 ? 2.f : c == 4 ? 1.f : 0.f)
           : r == 4 ? (c == 1 ? 2.f : c == 2 ? -1.f : c == 3 ? -2.f : c == 4 ? 1.f : 0.f)
                    : (c == 1 ? 4.f : c == 3 ? -5.f : c == 5 ? 1.f : 0.f);
}
__host__ __device__ constexpr float bt2(int r, int c) {          // B^T of F(2,3): 4 x 4
    return r == 0   ? (c == 0 ? 1.f : c == 2 ? -1.f : 0.f)
           : r == 1 ? (c == 1 ? 1.f : c == 2 ? 1.f : 0.f)
           : r == 2 ? (c == 1 ? -1.f : c == 2 ? 1.f : 0.f)
                    : (c == 1 ? 1.f : c == 3 ? -1.f : 0.f);
}
__host__ __device__ constexpr float at4(int r, int c) {          // A^T of F(4,3): 4 x 6
    return r == 0   ? (c < 5 ? 1.f : 0.f)
           : r == 1 ? (c == 1 ? 1.f : c == 2 ? -1.f : c == 3 ? 2.f : c == 4 ? -2.f : 0.f)
           : r == 2 ? (c == 1 ? 1.f : c == 2 ? 1.f : c == 3 ? 4.f : c == 4 ? 4.f : 0.f)
                    : (c == 1 ? 1.f : c == 2 ? -1.f : c == 3 ? 8.f : c == 4 ? -8.f : c == 5 ? 1.f : 0.f);
}
__host__ __device__ constexpr float at2(int r, int c) {          // A^T of F(2,3): 2 x 4
    return r == 0 ? (c < 3 ? 1.f : 0.f) : (c == 1 ? 1.f : c == 2 ? -1.f : c == 3 ? -1.f : 0.f);
}
__host__ __device__ constexpr float g4(int r, int c) {           // G of F(4,3): 6 x 3
    return r == 0   ? (c == 0 ? 0.25f : 0.f)
           : r == 1 ? (-1.f / 6.f)
           : r == 2 ? (c == 1 ? 1.f / 6.f : -1.f / 6.f)
           : r == 3 ? (c == 0 ? 1.f / 24.f : c == 1 ? 1.f / 12.f : 1.f / 6.f)
           : r == 4 ? (c == 0 ? 1.f / 24.f : c == 1 ? -1.f / 12.f : 1.f / 6.f)
                    : (c == 2 ? 1.f : 0.f);
}
__host__ __device__ constexpr float g2(int r, int c) {           // G of F(2,3): 4 x 3
    return r == 0 ? (c == 0 ? 1.f : 0.f) : r == 1 ? 0.5f : r == 2 ? (c == 1 ? -0.5f : 0.5f) : (c == 2 ? 1.f : 0.f);
}
template <int FY> __host__ __device__ constexpr float bty(int r, int c) { return FY == 2 ? bt2(r, c) : bt4(r, c); }
template <int FY> __host__ __device__ constexpr float aty(int r, int c) { return FY == 2 ? at2(r, c) : at4(r, c); }
__host__ __device__ constexpr float gy(int fy, int r, int c) { return fy == 2 ? g2(r, c) : g4(r, c); }

// offset / mask post-processing of SecondOrderDeformableAlignment (feat_prop.py:38-53), as conv.hip::dcn_post
__device__ __forceinline__ float w4_dcn_post(float v, int co, int C, const float* fl, float max_residue) {
    const int noff = (C / 3) * 2;
    if (co >= noff) return e2_fast_sigmoid(v);
    const int which = (co * 2 >= noff) ? 2 : 0;
    return max_residue * e2_fast_tanh(v) + fl[which + ((co & 1) ? 0 : 1)];
}
__device__ __forceinline__ float w4_act(float v, int act, float slope) {
    if (act == E2FGVI_ACT_RELU) return fmaxf(v, 0.f);
    if (act == E2FGVI_ACT_LRELU) return v > 0.f ? v : v * slope;
    if (act == E2FGVI_ACT_TANH) return tanhf(v);
    return v;
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t w4_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 w4_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}

constexpr int W4_RAW_W = 18;
constexpr unsigned W4_OOB = 0xFFFFFFFFu;
template <int V> struct IC4 { static constexpr int value = V; };
template <class F, int... Is>
__device__ __forceinline__ void w4_static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(IC4<Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void w4_static_for(F&& f) { w4_static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// pinned weight loads + explicit vmcnt, exactly as conv_wino.hip (the compiler would sink ordinary loads to their use)
typedef int w4_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ w4_i32x4 w4_rsrc_words(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    w4_i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xFFFFu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void w4_load4_pinned(f32x4& v, w4_i32x4 rsrc, unsigned byte_off) {
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(byte_off), "s"(rsrc) : "memory");
}
template <int N>
__device__ __forceinline__ void w4_wait_vmcnt() {
    // issued twice on purpose: the (free) duplicate marks this wait in the disassembly, where build.verify_wino_waits()
    // re-counts the vector-memory instructions between consecutive marked waits against N on every build
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// (Round 6 pruned what never won a layer -- profiles/r02_wino4_bench.txt: F(4x4,3x3) with its software-pipelined K loop, 12-wave
//  workgroups and 36 positions, and the 32-cout shape of F(2x4); the formulas below keep FY as a name, FY = 2 is what is built.)
template <int FY, int BN, int SC>
__global__ __launch_bounds__(128 * (FY + 2), 2) void conv_wino4_kernel(const W4Params p) {
    static_assert(FY == 2, "row tile: F(2x4,3x3)");
    static_assert(SC == 2, "chunks per LDS stage (the weight double buffer alternates with the chunk parity)");
    constexpr int NW = 2 * (FY + 2);
    constexpr int NT = 64 * NW;
    constexpr int TN = BN / 32;
    constexpr int NPOS = (FY + 2) * 6;
    constexpr int BH = 8 * FY;                                // block: BH x 16 pixels = 8 x 4 tiles of FY x 4
    constexpr int RAW_H = BH + 2;
    constexpr int PLANE_ROW = (FY == 2) ? 6 : 5;              // FY * PLANE_ROW = 4 or 12 (mod 16): conflict-free b128 patch reads
    constexpr int PLANE_RAW = RAW_H * PLANE_ROW * 16;
    constexpr int PLANE_BYTES = PLANE_RAW + ((16 - PLANE_RAW % 128) + 128) % 128;   // = 16 (mod 128): the staging stores
    static_assert(PLANE_BYTES % 128 == 16, "plane pitch");                          // of 8 lanes hit 8 distinct bank quads
    constexpr int CHUNK_BYTES = 8 * PLANE_BYTES;              // planes of one 8-channel chunk: [kq 2][column mod 4]
    constexpr int STAGE_BYTES = SC * CHUNK_BYTES;
    constexpr int EPI_BYTES = NPOS * 16 * 32 * 4;             // 16 tiles x 32 couts of every position
    constexpr int SMEM = (2 * STAGE_BYTES > EPI_BYTES) ? 2 * STAGE_BYTES : EPI_BYTES;
    constexpr int RAW_ITEMS = RAW_H * W4_RAW_W * 2;           // (pixel, kq) of one chunk
    constexpr int RAW_IT = (RAW_ITEMS + NT - 1) / NT;
    static_assert(NT >= 128 * FY, "epilogue items");

    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.y;
    const int logical = xcd_remap(blockIdx.x, p.nblk);
    const int mblocks = p.N * p.blocksY * p.blocksX;
    const int tile_n = logical / mblocks;
    int rem = logical - tile_n * mblocks;
    const int img = rem / (p.blocksY * p.blocksX);
    rem -= img * (p.blocksY * p.blocksX);
    const int by = rem / p.blocksX, bx = rem - by * p.blocksX;
    const int n0 = tile_n * BN;
    const int y0 = by * BH - 1, x0 = bx * 16 - 1;             // top-left of the raw patch

    // ---- raw-patch staging bookkeeping (chunk invariant)
    unsigned raw_off[RAW_IT];
    int raw_dst[RAW_IT];
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) {
        const int item = tid + it * NT;
        const int kq = item & 1, px = item >> 1;
        const int py = px / W4_RAW_W, pxx = px - py * W4_RAW_W;
        const int gy_ = y0 + py, gx_ = x0 + pxx;
        const bool have = item < RAW_ITEMS;
        const bool in = have && gy_ >= 0 && gy_ < p.H && gx_ >= 0 && gx_ < p.W;
        raw_off[it] = in ? (unsigned)((img * p.H + gy_) * p.W + gx_) : W4_OOB;
        raw_dst[it] = have ? (kq * 4 + (pxx & 3)) * PLANE_BYTES + (py * PLANE_ROW + (pxx >> 2)) * 16 : -1;
    }
    const unsigned raw_kq16 = (unsigned)(tid & 1) * 16u;

    int s = 0, c0 = 0;
    const float* cur_src = p.src[0];
    unsigned cur_bytes = p.src_bytes[0];
    unsigned cur_ld4 = (unsigned)p.ld[0] * 4u;
    unsigned cur_chan = (unsigned)(p.coff[0] + g * p.cpg[0]) * 4u;
    int cur_cpg = p.cpg[0];
    int raw_left = p.nchunks;       // chunks of the walk still inside the layer
    f32x4 rraw[SC][RAW_IT];
    auto load_raw = [&](f32x4 (&q)[RAW_IT]) {
        const __amdgpu_buffer_rsrc_t arsrc = w4_rsrc(cur_src, cur_bytes);
        const unsigned chan = cur_chan + (unsigned)c0 * 4u + raw_kq16;
        // chunks past the end (the prefetch runs a stage ahead of the K loop) fetch out of range: zeros, no memory access
        const bool cvalid = raw_left-- > 0 && c0 + (int)(raw_kq16 >> 2) < cur_cpg;
#pragma unroll
        for (int it = 0; it < RAW_IT; ++it) {
            // a select, never control flow: exactly ONE load per item on every path (the explicit vmcnt counts rely on it)
            unsigned off = (cvalid && raw_off[it] != W4_OOB) ? raw_off[it] * cur_ld4 + chan : W4_OOB;
            asm volatile("" : "+v"(off));
            q[it] = w4_load4(arsrc, off);
        }
        c0 += 8;
        if (c0 >= cur_cpg) {
            c0 = 0;
            ++s;
            if (s == p.nsrc) s = 0;
            if (p.nsrc > 1) {
                cur_src = p.src[s]; cur_bytes = p.src_bytes[s]; cur_ld4 = (unsigned)p.ld[s] * 4u;
                cur_chan = (unsigned)(p.coff[s] + g * p.cpg[s]) * 4u; cur_cpg = p.cpg[s];
            }
        }
    };
    auto store_raw = [&](int buf) {
        unsigned char* base = smem + buf * STAGE_BYTES;
#pragma unroll
        for (int q = 0; q < SC; ++q)
#pragma unroll
            for (int it = 0; it < RAW_IT; ++it)
                if (raw_dst[it] >= 0) *reinterpret_cast<f32x4*>(base + q * CHUNK_BYTES + raw_dst[it]) = rraw[q][it];
    };

    // ---- this wave's transform positions: row xi = wave >> 1, columns nu = 3 (wave & 1) + {0, 1, 2}
    const int xi = wave >> 1, half = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const int ty = i >> 2, tx = i & 3;
    // patch element (row r, column c) of the lane's tile, channel quad h: lane_base + (c & 3) planes + (r, c >> 2) units
    const int lane_base = (h * 4) * PLANE_BYTES + ((FY * ty) * PLANE_ROW + tx) * 16;

    // ---- B operands: lane (i, h) of position a, column tile n needs U[chunk][a][kq = h][n0 + 32 n + i][0..3]
    const w4_i32x4 wrsrc = w4_rsrc_words(p.w + (long long)g * p.wgroup_elems, p.wgroup_bytes);
    const unsigned u_step = (unsigned)(NPOS * 2) * (unsigned)p.Npad * 16u;          // bytes per chunk
    unsigned u_off[3][TN];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = n0 + n * 32 + i;
            const int pos = xi * 6 + 3 * half + a;
            u_off[a][n] = col < p.Npad ? (unsigned)(((pos * 2 + h) * p.Npad + col) * 16) : 0x80000000u;
        }
    f32x4 bq[2][3][TN];
    auto load_b = [&](int chunk, f32x4 (&q)[3][TN]) {
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int n = 0; n < TN; ++n) w4_load4_pinned(q[a][n], wrsrc, u_off[a][n] + (unsigned)chunk * u_step);
    };
    auto claim_b = [&](auto LATER_, f32x4 (&q)[3][TN]) {
        w4_wait_vmcnt<decltype(LATER_)::value>();
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int n = 0; n < TN; ++n) asm volatile("" : "+v"(q[a][n]));
    };

    f32x16 acc[3][TN];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][n][r] = 0.f;

    const int nstages = (p.nchunks + SC - 1) / SC;
#pragma unroll
    for (int q = 0; q < SC; ++q) load_raw(rraw[q]);
    load_b(0, bq[0]);
    store_raw(0);
#pragma unroll
    for (int q = 0; q < SC; ++q) load_raw(rraw[q]);
    __syncthreads();

    // The K loop is software-pipelined by one chunk: while the MFMAs of chunk c run, the same wave reads the patch of
    // chunk c+1 and builds its A operands (LDS reads + VALU issue in the shadow of the 64-cycle MFMAs).  Without this the
    // waves of a workgroup fall into lockstep -- all transforming, then all multiplying -- and the matrix pipe idles for
    // the length of a transform every chunk (measured: 67 % MFMA-busy against 82 % for the F(2x2) kernel, whose transform
    // is a smaller share of a chunk).  The next stage's patch is therefore parked in LDS in the MIDDLE of a stage (after
    // its first chunk), so that the last chunk of a stage can already read the first chunk of the next.
    auto k_loop = [&](auto XI_, auto HF_) {
        constexpr int XI = decltype(XI_)::value;
        constexpr int HF = decltype(HF_)::value;
        // patch reads of column HF + j: the rows with a non-zero B_y^T coefficient of the wave's xi
        auto read_col = [&](const unsigned char* raw, f32x4 (&d)[FY + 2], auto J_) {
            constexpr int col = HF + decltype(J_)::value;
#pragma unroll
            for (int r = 0; r < FY + 2; ++r)
                if (bty<FY>(XI, r) != 0.f)
                    d[r] = *reinterpret_cast<const f32x4*>(raw + (col & 3) * PLANE_BYTES + (r * PLANE_ROW + (col >> 2)) * 16);
        };
        // row combination of the wave's xi
        auto row_comb = [&](const f32x4 (&d)[FY + 2]) -> f32x4 {
            bool first = true;
            f32x4 acc_e = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < FY + 2; ++r) {
                const float c = bty<FY>(XI, r);
                if (c == 0.f) continue;
                acc_e = first ? c * d[r] : acc_e + c * d[r];
                first = false;
            }
            return acc_e;
        };
        // column combination nu = 3 HF + a
        auto col_comb = [&](const f32x4 (&e)[5], auto A_) -> f32x4 {
            constexpr int nu = 3 * HF + decltype(A_)::value;
            bool first = true;
            f32x4 acc_v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float c = bt4(nu, HF + j);
                if (c == 0.f) continue;
                acc_v = first ? c * e[j] : acc_v + c * e[j];
                first = false;
            }
            return acc_v;
        };
        // plain form: per chunk transform, then multiply (two workgroups per CU cover each other's transform phases)
        for (int st = 0; st < nstages; ++st) {
            const unsigned char* stage = smem + (st & 1) * STAGE_BYTES + lane_base;
#pragma unroll
            for (int q = 0; q < SC; ++q) {
                load_b(SC * st + q + 1, bq[(q & 1) ^ 1]);        // next chunk's weights land during this chunk
                if (q == 0) claim_b(IC4<3 * TN + SC * RAW_IT>{}, bq[q & 1]);
                else claim_b(IC4<3 * TN>{}, bq[q & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const unsigned char* raw = stage + q * CHUNK_BYTES;
                f32x4 d[5][FY + 2];
                f32x4 e[5];
                f32x4 v[3];
                w4_static_for<5>([&](auto J_) { read_col(raw, d[decltype(J_)::value], J_); });
                w4_static_for<5>([&](auto J_) { e[decltype(J_)::value] = row_comb(d[decltype(J_)::value]); });
                v[0] = col_comb(e, IC4<0>{}); v[1] = col_comb(e, IC4<1>{}); v[2] = col_comb(e, IC4<2>{});
#pragma unroll
                for (int k = 0; k < 4; ++k)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
#pragma unroll
                        for (int a = 0; a < 3; ++a)
                            acc[a][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[a][k], bq[q & 1][a][n][k], acc[a][n], 0, 0, 0);
            }
            store_raw((st & 1) ^ 1);
#pragma unroll
            for (int q = 0; q < SC; ++q) load_raw(rraw[q]);
            __syncthreads();
        }
        __syncthreads();                                          // the epilogue reuses the LDS
    };
    switch (wave) {          // wave-uniform
        case 0: k_loop(IC4<0>{}, IC4<0>{}); break;
        case 1: k_loop(IC4<0>{}, IC4<1>{}); break;
        case 2: k_loop(IC4<1>{}, IC4<0>{}); break;
        case 3: k_loop(IC4<1>{}, IC4<1>{}); break;
        case 4: k_loop(IC4<2>{}, IC4<0>{}); break;
        case 5: k_loop(IC4<2>{}, IC4<1>{}); break;
        case 6: k_loop(IC4<3>{}, IC4<0>{}); break;
        case 7: k_loop(IC4<3>{}, IC4<1>{}); break;
        default: break;
    }
    // The weight loads issued for the chunk PAST the end (out of range: zeros) are still in flight, and for hipcc an asm load's
    // destination is written when the statement ends: wait, then name the registers, so that nothing of the epilogue is allocated
    // on top of them before the data has landed (conv_wino.hip; checked by build.verify_exit_reuse()).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int n = 0; n < TN; ++n) E2_CLAIM_AFTER_LOOP(bq[b_][a][n]);

    // ---- epilogue: 16 tiles at a time: the positions meet in LDS, inverse transform, bias, residual, activation, store
    float* E = reinterpret_cast<float*>(smem);
    const int HW = p.H * p.W;
    const int pos0 = xi * 6 + 3 * half;
#pragma unroll
    for (int nh = 0; nh < TN; ++nh) {
#pragma unroll
        for (int th = 0; th < 2; ++th) {
            if (nh || th) __syncthreads();
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = th * 8 + rr;                                   // accumulator rows of tiles 16 th .. 16 th + 15
                    const int lt = (rr & 3) + 8 * (rr >> 2) + 4 * h;
                    E[((pos0 + a) * 16 + lt) * 32 + i] = acc[a][nh][r];
                }
            __syncthreads();
            // one (tile, output row, 4 consecutive couts) item per thread
            const int cq = tid & 7, lt = (tid >> 3) & 15, oy = tid >> 7;
            const int tile = 16 * th + lt;
            const int n = n0 + nh * 32 + cq * 4;
            const int oyy = by * BH + FY * (tile >> 2) + oy, oxx = bx * 16 + 4 * (tile & 3);
            // H is a multiple of FY and W of 4: a tile is inside or outside the image as a whole
            if (tid < 128 * FY && n < p.Cout_g && oyy < p.H && oxx < p.W) {
                f32x4 sc[6];
#pragma unroll
                for (int nu = 0; nu < 6; ++nu) sc[nu] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int x = 0; x < FY + 2; ++x) {
                    float cy = 0.f;                                              // A_y^T[oy][x], oy is a run-time value
#pragma unroll
                    for (int o = 0; o < FY; ++o) cy = (oy == o) ? aty<FY>(o, x) : cy;
#pragma unroll
                    for (int nu = 0; nu < 6; ++nu) {
                        const f32x4 m = *reinterpret_cast<const f32x4*>(E + ((x * 6 + nu) * 16 + lt) * 32 + cq * 4);
                        sc[nu] = sc[nu] + cy * m;
                    }
                }
                f32x4 y[4];
#pragma unroll
                for (int ox = 0; ox < 4; ++ox) {
                    bool first = true;
                    f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int nu = 0; nu < 6; ++nu) {
                        const float c = at4(ox, nu);
                        if (c == 0.f) continue;
                        a4 = first ? c * sc[nu] : a4 + c * sc[nu];
                        first = false;
                    }
                    y[ox] = a4;
                }
                const int co = g * p.Cout_g + n;
                const long long pix0 = (long long)img * HW + (long long)oyy * p.W + oxx;
                const bool full = n + 3 < p.Cout_g;
                f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                if (p.bias) {
#pragma unroll
                    for (int c = 0; c < 4; ++c) bv[c] = (full || n + c < p.Cout_g) ? p.bias[co + c] : 0.f;
                }
#pragma unroll
                for (int px = 0; px < 4; ++px) {
                    const long long pix = pix0 + px;
                    f32x4 v = y[px] + bv;
                    if (p.act == E2FGVI_ACT_DCNPOST) {
                        const f32x4 fl = *reinterpret_cast<const f32x4*>(p.res + pix * 4);
                        const float flv[4] = {fl[0], fl[1], fl[2], fl[3]};
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = w4_dcn_post(v[c], co + c, p.Cout, flv, p.slope);
                    } else {
                        if (p.res) {
                            const float* r = p.res + pix * p.res_ld + p.res_coff + co;
                            if (p.vec_store && full) v = v + *reinterpret_cast<const f32x4*>(r);
                            else {
#pragma unroll
                                for (int c = 0; c < 4; ++c) if (full || n + c < p.Cout_g) v[c] += r[c];
                            }
                        }
#pragma unroll
                        for (int c = 0; c < 4; ++c) v[c] = w4_act(v[c], p.act, p.slope);
                    }
                    float* o = p.dst + pix * p.dst_ld + p.dst_coff + co;
                    if (p.vec_store && full) *reinterpret_cast<f32x4*>(o) = v;
                    else {
#pragma unroll
                        for (int c = 0; c < 4; ++c) if (full || n + c < p.Cout_g) o[c] = v[c];
                    }
                }
            }
        }
    }
}

struct W4Pack {
    int Cout, groups, nsrc, fy, npos;
    int cpg[E2FGVI_MAX_SRC];
    int Cout_g, Npad, Cin_g, nchunks;
    long long wgroup_elems, total;
};

bool w4_geometry(int Cout, int groups, int nsrc, const int32_t* cpg, int fy, W4Pack* q) {
    if (Cout <= 0 || groups <= 0 || Cout % groups || nsrc < 1 || nsrc > E2FGVI_MAX_SRC || fy != 2) return false;
    q->Cout = Cout; q->groups = groups; q->nsrc = nsrc; q->fy = fy; q->npos = (fy + 2) * 6;
    q->Cout_g = Cout / groups;
    q->Npad = round_up(q->Cout_g, 32);
    q->Cin_g = 0; q->nchunks = 0;
    for (int s = 0; s < E2FGVI_MAX_SRC; ++s) q->cpg[s] = 0;
    for (int s = 0; s < nsrc; ++s) {
        if (cpg[s] <= 0 || cpg[s] % 4) return false;
        q->cpg[s] = cpg[s];
        q->Cin_g += cpg[s];
        q->nchunks += (cpg[s] + 7) / 8;
    }
    q->wgroup_elems = (long long)q->nchunks * q->npos * 2 * q->Npad * 4;
    q->total = q->wgroup_elems * groups;
    return true;
}

__global__ void pack_wino4_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, const W4Pack p) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.total) return;
    const int g = (int)(idx / p.wgroup_elems);
    long long rem = idx - (long long)g * p.wgroup_elems;
    const int kk = (int)(rem & 3);
    rem >>= 2;
    const int n = (int)(rem % p.Npad);
    rem /= p.Npad;
    const int kq = (int)(rem & 1);
    rem >>= 1;
    const int a = (int)(rem % p.npos);
    const int chunk = (int)(rem / p.npos);
    int s = 0, prefix = 0, lc = chunk;                     // chunk -> (source, chunk inside the source)
    while (lc >= (p.cpg[s] + 7) / 8) { lc -= (p.cpg[s] + 7) / 8; prefix += p.cpg[s]; ++s; }
    const int ch = lc * 8 + kq * 4 + kk;
    float v = 0.f;
    if (n < p.Cout_g && ch < p.cpg[s]) {
        const float* f = w + ((long long)(g * p.Cout_g + n) * p.Cin_g + prefix + ch) * 9;
        const int xi = a / 6, nu = a - xi * 6;
        // U = G_y f G_x^T
        float acc = 0.f;
        for (int r = 0; r < 3; ++r) {
            const float gr = gy(p.fy, xi, r);
            float row = 0.f;
            for (int c = 0; c < 3; ++c) row += f[r * 3 + c] * g4(nu, c);
            acc += gr * row;
        }
        v = acc;
    }
    wp[idx] = v;
}

template <int FY, int BN>
int launch_wino4(W4Params& p, int groups, hipStream_t st) {
    p.blocksY = cdiv(p.H, 8 * FY);
    p.blocksX = cdiv(p.W, 16);
    p.tilesN = cdiv(p.Cout_g, BN);
    const long long nblk = (long long)p.N * p.blocksY * p.blocksX * p.tilesN;
    E2_REQUIRE(nblk < 2147483647LL, E2FGVI_EUNSUP, "conv3x3_winograd4: grid too large");
    p.nblk = (int)nblk;
    hipLaunchKernelGGL((conv_wino4_kernel<FY, BN, 2>), dim3(p.nblk, groups, 1), dim3(128 * (FY + 2)), 0, st, p);
    E2_LAUNCH_CHECK("conv3x3_winograd4");
    return 0;
}

}  // namespace

extern "C" int64_t e2fgvi_packed_winograd4_weight_size(int32_t Cout, int32_t groups, int32_t nsrc, const int32_t* src_cpg,
                                                       int32_t fy) {
    W4Pack q;
    if (!src_cpg || !w4_geometry(Cout, groups, nsrc, src_cpg, fy, &q)) {
        e2fgvi_set_error("packed_winograd4_weight_size: bad geometry (channels per source multiples of 4, fy = 2)");
        return E2FGVI_EINVAL;
    }
    return q.total;
}

extern "C" int e2fgvi_pack_winograd4_weight(const float* w, float* wpacked, int32_t Cout, int32_t groups, int32_t nsrc,
                                            const int32_t* src_cpg, int32_t fy, void* stream) {
    W4Pack q;
    E2_REQUIRE(w && wpacked && src_cpg, E2FGVI_EINVAL, "pack_winograd4_weight: null pointer");
    E2_REQUIRE(w4_geometry(Cout, groups, nsrc, src_cpg, fy, &q), E2FGVI_EINVAL, "pack_winograd4_weight: bad geometry");
    hipLaunchKernelGGL(pack_wino4_weight_kernel, dim3((unsigned)cdiv64(q.total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       wpacked, q);
    E2_LAUNCH_CHECK("pack_winograd4_weight");
    return 0;
}

extern "C" int e2fgvi_conv3x3_winograd4(const e2fgvi_conv_desc* d, int32_t fy, void* stream) {
    E2_REQUIRE(d, E2FGVI_EINVAL, "conv3x3_winograd4: null descriptor");
    W4Pack q;
    E2_REQUIRE(w4_geometry(d->Cout, d->groups, d->nsrc, d->src_cpg, fy, &q), E2FGVI_EINVAL,
               "conv3x3_winograd4: bad geometry (channels per source must be multiples of 4, fy = 2)");
    E2_REQUIRE(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1, E2FGVI_EUNSUP,
               "conv3x3_winograd4: only 3x3, stride 1, pad 1");
    E2_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->H % fy == 0 && d->W % 4 == 0, E2FGVI_EUNSUP,
               "conv3x3_winograd4: H must be a multiple of fy and W a multiple of 4");
    E2_REQUIRE(d->Ho == d->H && d->Wo == d->W, E2FGVI_EINVAL, "conv3x3_winograd4: Ho/Wo must equal H/W");
    E2_REQUIRE(d->wpacked && d->dst, E2FGVI_EINVAL, "conv3x3_winograd4: null weight/dst");
    E2_REQUIRE(!d->dst_nchw, E2FGVI_EUNSUP, "conv3x3_winograd4: NCHW output not supported");
    E2_REQUIRE(d->dst_coff >= 0 && d->dst_coff + d->Cout <= d->dst_ld, E2FGVI_EINVAL, "conv3x3_winograd4: dst slice exceeds dst_ld");
    if (d->act == E2FGVI_ACT_DCNPOST)
        E2_REQUIRE(d->residual && d->Cout % 3 == 0 && d->groups == 1 && ((uintptr_t)d->residual & 15) == 0, E2FGVI_EINVAL,
                   "conv3x3_winograd4: ACT_DCNPOST needs the [pixel][4] flows as residual, Cout %% 3 == 0, groups == 1");
    W4Params p;
    for (int s = 0; s < E2FGVI_MAX_SRC; ++s) { p.src[s] = nullptr; p.ld[s] = 0; p.coff[s] = 0; p.cpg[s] = 0; p.src_bytes[s] = 0; }
    for (int s = 0; s < d->nsrc; ++s) {
        E2_REQUIRE(d->src[s], E2FGVI_EINVAL, "conv3x3_winograd4: null source %d", s);
        E2_REQUIRE(d->src_ld[s] % 4 == 0 && d->src_coff[s] % 4 == 0 && ((uintptr_t)d->src[s] & 15) == 0, E2FGVI_EINVAL,
                   "conv3x3_winograd4: source %d not 16-byte addressable", s);
        E2_REQUIRE(d->src_coff[s] + d->groups * d->src_cpg[s] <= d->src_ld[s], E2FGVI_EINVAL,
                   "conv3x3_winograd4: source %d channel range exceeds its pixel stride", s);
        const long long bytes = (long long)d->N * d->H * d->W * d->src_ld[s] * 4;
        E2_REQUIRE(bytes < 4294967295LL, E2FGVI_EUNSUP, "conv3x3_winograd4: source %d spans >= 4 GiB (split the batch)", s);
        p.src[s] = d->src[s]; p.ld[s] = d->src_ld[s]; p.coff[s] = d->src_coff[s]; p.cpg[s] = d->src_cpg[s];
        p.src_bytes[s] = (unsigned)bytes;
    }
    E2_REQUIRE(q.wgroup_elems * 4 < 0x70000000LL, E2FGVI_EUNSUP, "conv3x3_winograd4: packed weight group >= 1.75 GiB");
    E2_REQUIRE(((uintptr_t)d->wpacked & 15) == 0, E2FGVI_EINVAL, "conv3x3_winograd4: packed weight not 16-byte aligned");
    p.nsrc = d->nsrc;
    p.N = d->N; p.H = d->H; p.W = d->W;
    p.Cout = d->Cout; p.Cout_g = q.Cout_g; p.Npad = q.Npad;
    p.nchunks = q.nchunks;
    p.wgroup_elems = q.wgroup_elems; p.wgroup_bytes = (unsigned)(q.wgroup_elems * 4);
    p.w = (const float*)d->wpacked; p.bias = d->bias;
    p.res = d->residual; p.res_ld = d->res_ld; p.res_coff = d->res_coff;
    p.dst = d->dst; p.dst_ld = d->dst_ld; p.dst_coff = d->dst_coff;
    p.act = d->act; p.slope = d->slope;
    bool vec = ((uintptr_t)d->dst & 15) == 0 && d->dst_ld % 4 == 0 && d->dst_coff % 4 == 0 && q.Cout_g % 4 == 0;
    if (d->residual && d->act != E2FGVI_ACT_DCNPOST)
        vec = vec && ((uintptr_t)d->residual & 15) == 0 && d->res_ld % 4 == 0 && d->res_coff % 4 == 0;
    p.vec_store = vec ? 1 : 0;
    hipStream_t st = (hipStream_t)stream;
    int tile = d->tile;
    if (!tile) tile = 64;
    if (fy == 2 && tile == 64) return launch_wino4<2, 64>(p, d->groups, st);
    e2fgvi_set_error("conv3x3_winograd4: fy must be 2 and tile 0 (auto) or 64 couts per workgroup (the other shapes were pruned in round 6)");
    return E2FGVI_EUNSUP;
}
