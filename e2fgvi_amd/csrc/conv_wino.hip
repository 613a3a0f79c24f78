// Winograd F(2x2, 3x3) fp32 convolution on the fp32 MFMA pipe (gfx950): 3x3, stride 1, pad 1, NHWC, virtual concat
// of up to 4 sources, groups.  16 multiplies per 2x2 outputs instead of 36 -> 2.25x less MFMA work than the implicit
// GEMM of conv.hip for the wide stride-1 layers (encoder, decoder, SoftComp bias conv).  Arithmetic stays fp32.
//
//   Y = A^T [ (G g G^T) (.) (B^T d B) ] A      per 4x4 input patch d, 3x3 filter g  (Lavin & Gray, F(2x2,3x3))
//   as 16 independent GEMMs  M_a[tile, cout] = sum_cin V_a[tile, cin] * U_a[cin, cout],  a = (xi, nu) in 4x4
//
// Workgroup = 8 waves = one 16x16-pixel output block (8x8 Winograd tiles = 2 MFMA row-tiles; 8x16 pixels / 1 row-tile
// in the small-image variant that the 60x108 propagation convs use) x BN output channels.
// Wave w owns the two transform positions a = 2w, 2w+1 (same xi = w>>1, nu in {0,1} or {2,3}) and keeps
// 2 x 2 x (BN/32) 32x32 accumulators.  Per 8-channel chunk of the input:
//   * the raw 18x18-pixel halo patch is staged ONCE in LDS (global -> registers -> LDS one stage ahead; a stage = two
//     chunks, so one barrier per 16 channels; even/odd pixel columns live in separate planes, which makes the stride-2
//     patch reads of the 32 tiles of an MFMA row-tile bank-conflict free);
//   * every lane builds its MFMA A operands straight from the patch: 6 ds_read_b128 (2 rows x 3 columns of the 4x4
//     patch, 4 channels each) and ~36 VALU ops give V for both of the wave's positions -- the transformed input never
//     exists in memory;
//   * B operands (the pre-transformed weights of the wave's own two positions) never touch LDS: one 16-byte buffer load
//     per (position, column tile), issued one chunk ahead, is exactly the lane's operand quad; 8 MFMAs per tile pair.
// Epilogue: the 16 M_a tiles meet in LDS, every thread applies A^T M A to one (tile, 4 couts) item, adds the bias and
// the residual, applies the activation (or the DCN offset/mask post-processing) and stores 2x2 pixels x 16 bytes.
//
// Packed weights: [group][chunk][a = 16][kq = 2][Npad][4]  (chunk = 8 input channels in concat order, sources padded
// to 8 -- a source may be 4 (mod 8) wide, its last half chunk is zero; kq = channel quad; Npad = Cout_g rounded up to
// 32), produced by e2fgvi_pack_winograd_weight.
//
// X3 = true (round 3; built as a second object, conv_wino_x3.o, without packed-fp32 VALU): the 16 GEMMs run on the bf16
// matrix pipe with EXACTLY split operands, as conv_bf16x.hip's MODE 2 does for the implicit GEMM.  The transformed input V
// (fp32, built in registers as before) is split into three bf16 numbers per value (8 + 8 + 8 significand bits, their sum is
// V bit for bit), the transformed weights U are split the same way at packing time, and of the nine bf16 products of a
// V * U pair the six largest are accumulated by v_mfma_f32_32x32x16_bf16 (the dropped ones are < 2^-22 of the product):
// fp32-level rounding for 6 x 32 MFMA cycles per 16 input channels instead of 8 x 64.  A lane then covers 8 channels of its
// tile (both channel quads of chunk h of the 16-channel LDS stage) instead of 4, one K iteration = one LDS stage;
// staging, epilogue and tile shapes are shared with the fp32 kernel.
//   X3 packed weights: [group][stage = 16 channels][a = 16][plane hi, mid, lo][h = chunk of the stage][Npad][8 bf16]
#include "common.h"
#include <stdlib.h>

#ifndef E2_WINO_X3
#define E2_WINO_X3 0           // 1: the split-bf16 build of this file (conv_wino_x3.o): only the X3 kernels and their entry points
#endif

// The register claims behind the K loops: an empty asm that names a weight register as read and written, placed
// behind the post-loop s_waitcnt vmcnt(0) -- the register stays allocated until the load that was in flight has landed.
// tests/test_host_logic.py compiles this file once with the macro defined empty and expects build.check_exit_reuse() to reject it.
#ifndef E2_CLAIM_AFTER_LOOP
#define E2_CLAIM_AFTER_LOOP(r) asm volatile("" : "+v"(r))
#endif

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// x = hi + mid + lo with bf16 pieces: common.h, e2_split2 / e2_split8
__device__ __forceinline__ void wino_split8(const f32x4& v0, const f32x4& v1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    e2_split8(v0, v1, hi, mid, lo);
}

struct WinoParams {
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
    int vec_store;          // destination (and residual) rows are 16-byte addressable for every group / cout tile
};

// offset / mask post-processing of SecondOrderDeformableAlignment fused into conv_offset's last layer -- same rule as
// conv.hip::dcn_post (feat_prop.py:38-53)
__device__ __forceinline__ float wino_dcn_post(float v, int co, int C, const float* fl, float max_residue) {
    const int noff = (C / 3) * 2;
    if (co >= noff) return e2_fast_sigmoid(v);
    const int which = (co * 2 >= noff) ? 2 : 0;
    return max_residue * e2_fast_tanh(v) + fl[which + ((co & 1) ? 0 : 1)];
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}

constexpr int RAW_W = 18;                 // raw patch is (8 MT + 2) x 18 pixels
constexpr int PLANE_ROW = 12;             // 16-byte units per patch row in one column-parity plane (9 used)
constexpr unsigned OOB = 0xFFFFFFFFu;
template <int V> struct IC { static constexpr int value = V; };

// The weight (B operand) loads must be ISSUED a whole chunk before their use; the compiler sinks ordinary loads down to
// just before the first MFMA that needs them and the L2 latency lands on the critical path (measured: ~1300 cycles per
// chunk).  They are therefore issued through inline asm, pinned in place, and waited for with an explicit s_waitcnt
// whose count is the number of vector loads issued after them.
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 rsrc_words(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xFFFFu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void buf_load4_pinned(f32x4& v, i32x4 rsrc, unsigned byte_off) {
    // "memory": the compiler's own loads (the patch prefetch) must keep their program order around these, the explicit
    // vmcnt counts below depend on it
    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(v) : "v"(byte_off), "s"(rsrc) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    // issued twice on purpose: the (free) duplicate marks this wait in the disassembly, where build.verify_wino_waits()
    // re-counts the vector-memory instructions between consecutive marked waits against N on every build
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// One LDS-DMA piece of the raw patch (DMA variant below): 64 lanes x 16 bytes of the buffer `rsrc` at the lanes' byte offsets
// (out-of-range offsets land as zeros) -> LDS bytes [lds_dst, lds_dst + 1024), lds_dst wave-uniform.  Inline asm like the
// weight loads: hipcc counts neither the load nor any wait for it, the kernel's explicit vmcnt waits cover it, and
// build.verify_wino_waits() counts it in the disassembly like every other vector-memory instruction.  M0 is written in the
// statement that reads it and restored (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ void dma_piece(i32x4 rsrc, unsigned lds_dst, unsigned voff) {
    // (s_nop 2: five wait states between a v_readfirstlane / v_readlane that produced the resource words and the load that reads them)
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 2\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
typedef __attribute__((address_space(3))) void wino_lds_void;

// MT = MFMA row-tiles per workgroup: 2 -> 16x16-pixel block (the big layers), 1 -> 8x16 (small images: more workgroups)
// (Round 3's LDS-DMA staging of the patch with two stages of look-ahead -- tile codes + 1000 -- measured slower on every layer
//  in both arithmetics, profiles/r03_wino_dma_staging.txt, and was removed in round 6; the wide-tile kernel below keeps its own.)
template <int MT, int BN, int SC, bool X3>
__global__ __launch_bounds__(512, ((MT == 1 && BN == 32 && !X3) ? 4 : 2)) void conv_wino_kernel(const WinoParams p) {
    static_assert(SC == 2, "chunks per LDS stage (the explicit vmcnt counts of k_loop assume two)");
    constexpr int NT = 512;
    constexpr int TN = BN / 32;
    constexpr int RAW_H = 8 * MT + 2;
    constexpr int PLANE_RAW = RAW_H * PLANE_ROW * 16;
    // plane pitch = 16 (mod 128): the 8 lanes of a staging store (4 pixels x 2 channel quads = the 4 planes twice) hit 8
    // distinct bank quads; with a pitch of 0 (mod 128) the planes collided 4-way (SQ_LDS_BANK_CONFLICT was 32 % of the LDS
    // cycles, profiles/r02_wino4_pmc_enc10.txt)
    constexpr int PLANE_BYTES = PLANE_RAW + ((16 - PLANE_RAW % 128) + 128) % 128;
    constexpr int CHUNK_USED = 4 * PLANE_BYTES;               // planes of one 8-channel chunk: [kq][column parity]
    constexpr int CHUNK_BYTES = CHUNK_USED;
    constexpr int STAGE_BYTES = SC * CHUNK_BYTES;             // an LDS stage holds SC chunks (one barrier per 8 SC channels)
    constexpr int NSTAGE = 2;
    constexpr int TILES = 32 * MT;
    constexpr int EPI_BYTES = 16 * TILES * 32 * 4;
    constexpr int SMEM = (NSTAGE * STAGE_BYTES > EPI_BYTES) ? NSTAGE * STAGE_BYTES : EPI_BYTES;
    constexpr int RAW_ITEMS = RAW_H * RAW_W * 2;          // (pixel, kq) of one chunk
    constexpr int RAW_IT = (RAW_ITEMS + NT - 1) / NT;

    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int g = blockIdx.y;
    const int logical = xcd_remap(blockIdx.x, p.nblk);
    // spatial block fastest: the CUs of one XCD share the weight slab of one cout tile in their L2
    const int mblocks = p.N * p.blocksY * p.blocksX;
    const int tile_n = logical / mblocks;
    int rem = logical - tile_n * mblocks;
    const int img = rem / (p.blocksY * p.blocksX);
    rem -= img * (p.blocksY * p.blocksX);
    const int by = rem / p.blocksX, bx = rem - by * p.blocksX;
    const int n0 = tile_n * BN;
    const int y0 = by * (8 * MT) - 1, x0 = bx * 16 - 1;    // top-left of the raw patch

    // ---- raw-patch staging bookkeeping (chunk invariant)
    unsigned raw_off[RAW_IT];      // pixel index in its source, OOB if outside the image / no item
    int raw_dst[RAW_IT];           // LDS byte offset inside a chunk area, -1 if the thread has no item
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) {
        const int item = tid + it * NT;
        const int kq = item & 1, px = item >> 1;
        const int py = px / RAW_W, pxx = px - py * RAW_W;
        const int gy = y0 + py, gx = x0 + pxx;
        const bool have = item < RAW_ITEMS;
        const bool in = have && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        raw_off[it] = in ? (unsigned)((img * p.H + gy) * p.W + gx) : OOB;
        raw_dst[it] = have ? (kq * 2 + (pxx & 1)) * PLANE_BYTES + (py * PLANE_ROW + (pxx >> 1)) * 16 : -1;
    }
    const unsigned raw_kq16 = (unsigned)(tid & 1) * 16u;            // NT is even: the kq of an item is the thread's parity

    // Chunks past the end (odd chunk count, prefetch overrun) need no guards: their weight loads fall outside the
    // group's buffer range and return zeros, so whatever patch data is re-read contributes nothing.
    // The parameters of the source being walked live in (scalar) registers and are re-read from the kernel arguments only
    // when the walk crosses into the next source -- indexing p.src[s] / p.ld[s] / ... per chunk would put two dependent
    // scalar-memory round trips in front of every chunk's patch loads.
    int s = 0, c0 = 0;              // next chunk to load: source, first channel inside the group's slice
    const float* cur_src = p.src[0];
    unsigned cur_bytes = p.src_bytes[0];
    unsigned cur_ld4 = (unsigned)p.ld[0] * 4u;
    unsigned cur_chan = (unsigned)(p.coff[0] + g * p.cpg[0]) * 4u;
    int cur_cpg = p.cpg[0];
    int raw_left = p.nchunks;       // chunks of the walk still inside the layer (see load_raw)
    f32x4 rraw[SC][RAW_IT];
    auto load_raw = [&](f32x4 (&q)[RAW_IT]) {
        const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(cur_src, cur_bytes);
        const unsigned chan = cur_chan + (unsigned)c0 * 4u + raw_kq16;
        // a source may end in the middle of a chunk (cpg % 8 == 4); chunks past the end (the prefetch runs a stage ahead of the K
        // loop) fetch out of range: zeros, returned without a memory access -- the wait behind the K loop then costs nothing and
        // the first chunks are not re-read for nothing
        const bool cvalid = raw_left-- > 0 && c0 + (int)(raw_kq16 >> 2) < cur_cpg;
#pragma unroll
        for (int it = 0; it < RAW_IT; ++it) {
            // a select, never control flow: exactly ONE load per item on every path (the explicit vmcnt counts rely on
            // it; with a branch per condition the compiler issued one load per arm)
            unsigned off = (cvalid && raw_off[it] != OOB) ? raw_off[it] * cur_ld4 + chan : OOB;
            asm volatile("" : "+v"(off));
            q[it] = buf_load4(arsrc, off);
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

    const unsigned smem_lds = (unsigned)(unsigned long long)(wino_lds_void*)smem;
    const float* const dsp0 = p.src[0]; const float* const dsp1 = p.src[1]; const float* const dsp2 = p.src[2]; const float* const dsp3 = p.src[3];
    const unsigned dsb0 = p.src_bytes[0], dsb1 = p.src_bytes[1], dsb2 = p.src_bytes[2], dsb3 = p.src_bytes[3];
    const unsigned dsl0 = (unsigned)p.ld[0] * 4u, dsl1 = (unsigned)p.ld[1] * 4u, dsl2 = (unsigned)p.ld[2] * 4u, dsl3 = (unsigned)p.ld[3] * 4u;
    const unsigned dsc0 = (unsigned)(p.coff[0] + g * p.cpg[0]) * 4u, dsc1 = (unsigned)(p.coff[1] + g * p.cpg[1]) * 4u,
                   dsc2 = (unsigned)(p.coff[2] + g * p.cpg[2]) * 4u, dsc3 = (unsigned)(p.coff[3] + g * p.cpg[3]) * 4u;
    const int dsg0 = p.cpg[0], dsg1 = p.cpg[1], dsg2 = p.cpg[2], dsg3 = p.cpg[3];

    // ---- this wave's transform positions: xi = wave >> 1, nu in {0,1} (pair A) or {2,3} (pair B)
    //   B^T rows:  0: d0 - d2   1: d1 + d2   2: -d1 + d2   3: d1 - d3          (same combinations over columns)
    //   pair A reads patch columns 0,1,2:  nu0 = c0 - c2,  nu1 = c1 + c2
    //   pair B reads patch columns 1,2,3:  nu2 = c2 - c1,  nu3 = c1 - c3
    const int xi = wave >> 1;
    const int ra0 = (xi == 0) ? 0 : 1, ra1 = (xi == 3) ? 3 : 2;
    const int cb = wave & 1;
    const int i = lane & 31, h = lane >> 5;
    const int ty = i >> 3, tx = i & 7;
    // LDS byte offsets (inside a chunk area) of the 6 patch reads for row-tile 0; row-tile 1 is 8 patch rows lower
    int a_off[2][3];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int prow = 2 * ty + (r ? ra1 : ra0);
            const int col = cb + j;                                  // patch column 0..3 -> pixel column 2*tx + col
            // fp32: the lane's channel quad kq = h of every chunk.  X3: both quads (+ 2 PLANE_BYTES for kq = 1) of chunk h
            a_off[r][j] = X3 ? h * CHUNK_BYTES + (col & 1) * PLANE_BYTES + (prow * PLANE_ROW + tx + (col >> 1)) * 16
                             : (h * 2 + (col & 1)) * PLANE_BYTES + (prow * PLANE_ROW + tx + (col >> 1)) * 16;
        }

    // ---- B operands (pre-transformed weights) go global -> registers, no LDS: lane (i, h) of position a, column
    // tile n needs U[chunk][2*wave + a][kq = h][n0 + 32 n + i][0..3] = one 16-byte load
    const i32x4 wrsrc = rsrc_words(reinterpret_cast<const char*>(p.w) + (long long)g * p.wgroup_bytes, p.wgroup_bytes);
    const unsigned u_step = (X3 ? 96u : 32u) * (unsigned)p.Npad * 16u;   // bytes per chunk (X3: per 16-channel stage, 3 planes)
    unsigned u_off[2][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = n0 + n * 32 + i;
            // columns past Npad: an offset that stays out of range for every chunk (group size is < 0x70000000 bytes)
            // X3: plane pl adds 2 Npad 16-byte units ([a][plane][h][Npad])
            u_off[a][n] = col < p.Npad ? (unsigned)(((X3 ? (2 * wave + a) * 6 + h : (2 * wave + a) * 2 + h) * p.Npad + col) * 16)
                                       : 0x80000000u;
        }
    const unsigned u_plane = 2u * (unsigned)p.Npad * 16u;
    f32x4 bq[2][2][TN];
    auto load_b = [&](int chunk, f32x4 (&q)[2][TN]) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int n = 0; n < TN; ++n) buf_load4_pinned(q[a][n], wrsrc, u_off[a][n] + (unsigned)chunk * u_step);
    };
    // X3: the weights of a 16-channel stage, 3 planes per (position, column tile)
    f32x4 bw[2][2][X3 ? TN : 1][3];
    auto load_b3 = [&](int st, f32x4 (&q)[2][X3 ? TN : 1][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int n = 0; n < (X3 ? TN : 1); ++n)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    buf_load4_pinned(q[a][n][pl], wrsrc, u_off[a][n] + (unsigned)pl * u_plane + (unsigned)st * u_step);
    };
    // the chunk's weights were issued one chunk ago; LATER is the number of vector loads issued since then
    auto claim_b = [&](auto LATER_, f32x4 (&q)[2][TN]) {
        wait_vmcnt<decltype(LATER_)::value>();
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int n = 0; n < TN; ++n) asm volatile("" : "+v"(q[a][n]));
    };

    f32x16 acc[2][MT][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][m][n][r] = 0.f;

    const int nstages = (p.nchunks + SC - 1) / SC;
    {
#pragma unroll
        for (int q = 0; q < SC; ++q) load_raw(rraw[q]);
        if constexpr (X3) load_b3(0, bw[0]);
        else load_b(0, bq[0]);
        store_raw(0);
#pragma unroll
        for (int q = 0; q < SC; ++q) load_raw(rraw[q]);
        __syncthreads();
    }

    // the K loop, specialised on the wave's role so that the input transform is plain adds / subtracts
    auto k_loop = [&](auto XI_, auto PB_) __attribute__((always_inline)) {
        constexpr int XI = decltype(XI_)::value;
        constexpr bool PB = decltype(PB_)::value != 0;
        // Waves w and w + 4 share a SIMD.  Measured with tools/wino_timing.py (profiles/r02_wino_timing.txt): per stage a wave
        // spends ~10 % of its K-loop cycles parking the next stage's patch and issuing the loads of the stage after, and with
        // every wave doing that right before the stage barrier the matrix pipe idled meanwhile.  The upper four waves (xi >= 2)
        // therefore do it in the MIDDLE of the stage, under the MFMAs of their SIMD partners, and vice versa.  The target
        // buffer is free for the whole stage (its readers passed the previous barrier) and the stage barrier still follows
        // every wave's stores.
        constexpr bool LATE = false;      // (mid-stage parking of the upper waves: measured neutral, profiles/r02_wino_timing.txt)
        if constexpr (X3) {
            // one iteration per LDS stage (16 channels): the stage's weights (2 positions x TN column tiles x 3 planes) were
            // issued a stage ago; since then: the patch prefetch of the stage after (SC * RAW_IT loads) and the next weights
            // one stage's arithmetic: 12 patch reads, the transform of 8 channels for both positions, two splits, 12 TN MFMAs
            // ... in two halves, so that the register-staged loop can run the MFMAs of stage st under the reads / transform / split
            // of stage st + 1: x3_prep builds the six operand fragments of a stage, x3_mma issues the 12 TN MFMAs on them
            auto x3_prep = [&](const unsigned char* stage, bf16x8 (&A)[MT][6]) __attribute__((always_inline)) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const unsigned char* rm = stage + m * (8 * PLANE_ROW * 16);
                    f32x4 va[2], vb[2];
#pragma unroll
                    for (int kq = 0; kq < 2; ++kq) {
                        f32x4 e[3];
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const f32x4 d0 = *reinterpret_cast<const f32x4*>(rm + kq * (2 * PLANE_BYTES) + a_off[0][j]);
                            const f32x4 d1 = *reinterpret_cast<const f32x4*>(rm + kq * (2 * PLANE_BYTES) + a_off[1][j]);
                            e[j] = XI == 1 ? d0 + d1 : XI == 2 ? d1 - d0 : d0 - d1;
                        }
                        va[kq] = PB ? e[1] - e[0] : e[0] - e[2];
                        vb[kq] = PB ? e[0] - e[2] : e[1] + e[2];
                    }
                    wino_split8(va[0], va[1], A[m][0], A[m][1], A[m][2]);
                    wino_split8(vb[0], vb[1], A[m][3], A[m][4], A[m][5]);
                }
            };
            auto x3_mma = [&](const bf16x8 (&A)[MT][6], f32x4 (&wq)[2][TN][3]) __attribute__((always_inline)) {
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const bf16x8 ah = A[m][0], am = A[m][1], al = A[m][2], bh = A[m][3], bm = A[m][4], bl = A[m][5];
#pragma unroll
                    for (int n = 0; n < TN; ++n) {
                        const bf16x8 u0h = __builtin_bit_cast(bf16x8, wq[0][n][0]), u0m = __builtin_bit_cast(bf16x8, wq[0][n][1]),
                                     u0l = __builtin_bit_cast(bf16x8, wq[0][n][2]);
                        const bf16x8 u1h = __builtin_bit_cast(bf16x8, wq[1][n][0]), u1m = __builtin_bit_cast(bf16x8, wq[1][n][1]),
                                     u1l = __builtin_bit_cast(bf16x8, wq[1][n][2]);
                        // smallest terms first, the two positions interleaved (independent accumulators)
                        acc[0][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, u0h, acc[0][m][n], 0, 0, 0);
                        acc[1][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl, u1h, acc[1][m][n], 0, 0, 0);
                        acc[0][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, u0l, acc[0][m][n], 0, 0, 0);
                        acc[1][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, u1l, acc[1][m][n], 0, 0, 0);
                        acc[0][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, u0m, acc[0][m][n], 0, 0, 0);
                        acc[1][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm, u1m, acc[1][m][n], 0, 0, 0);
                        acc[0][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, u0h, acc[0][m][n], 0, 0, 0);
                        acc[1][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bm, u1h, acc[1][m][n], 0, 0, 0);
                        acc[0][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, u0m, acc[0][m][n], 0, 0, 0);
                        acc[1][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, u1m, acc[1][m][n], 0, 0, 0);
                        acc[0][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, u0h, acc[0][m][n], 0, 0, 0);
                        acc[1][m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh, u1h, acc[1][m][n], 0, 0, 0);
                    }
                }
            };
            auto claim3 = [&](f32x4 (&wq)[2][TN][3]) __attribute__((always_inline)) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(wq[a][n][pl]));
            };
            
            // Register-staged patch (the product variant).  Entry state (prologue above): buffer 0 = stage 0 (visible), staging
            // registers = stage 1, stage 0's weights issued between the two patch loads -- like the fp32 kernel's.
            // (two stages per trip with compile-time buffer indices: `bw[st & 1]` would put the weight registers in scratch; an odd
            //  stage count runs one stage past the end -- zero weights, like the fp32 kernel's chunk overrun)
            // Measured and dropped (profiles/r03_x3_winograd_variants.txt): building the fragments of stage st + 1 under the MFMAs
            // of stage st (one more fragment set, the same loads) ran 0-8 % slower, with the compiler's own instruction order and
            // with an explicit one-MFMA-per-16-VALU interleave (sched_group_barrier) alike; the LDS-DMA variant above 10-40 % slower;
            // weights fetched two stages ahead into a third buffer with the patch loads pinned (so that the compiler's vmcnt(0) in
            // front of store_raw no longer drains the next stage's weights): 3-8 % slower; all eight waves on ONE role's code (an
            // instruction-cache test, wrong results): identical time.  What bounds the ~1.1 us per 16-channel stage is none of:
            // matrix pipe, phase serialisation inside a SIMD, weight / patch latency, instruction fetch.
            auto stage_body = [&](auto CUR_, int st) __attribute__((always_inline)) {
                constexpr int CUR = decltype(CUR_)::value;
                load_b3(st + 1, bw[CUR ^ 1]);
                wait_vmcnt<6 * TN + SC * RAW_IT>();
                claim3(bw[CUR]);
                __builtin_amdgcn_sched_barrier(0);
                {
                    bf16x8 A[MT][6];
                    x3_prep(smem + CUR * STAGE_BYTES, A);
                    // (timing build only: fragments complete before the first MFMA issues, so that the two segments separate)
                    x3_mma(A, bw[CUR]);
                }
                // the registers hold stage st + 1: park it in the other buffer (its readers passed the previous barrier)
                store_raw(CUR ^ 1);
#pragma unroll
                for (int q2 = 0; q2 < SC; ++q2) load_raw(rraw[q2]);
                __syncthreads();
            };
            for (int st = 0; st < nstages; st += 2) {
                stage_body(IC<0>{}, st);
                stage_body(IC<1>{}, st + 1);
            }
            return;
        }
        
        for (int st = 0; st < nstages; ++st) {
            const unsigned char* stage = smem + (st & 1) * STAGE_BYTES;
#pragma unroll
            for (int q = 0; q < SC; ++q) {
                load_b(SC * st + q + 1, bq[(q & 1) ^ 1]);            // next chunk's weights land during this chunk
                // loads issued after this chunk's weights: the next chunk's (2 TN) and, across a stage boundary, the
                // patch prefetch (SC * RAW_IT)
                // (upper waves: the patch prefetch sits between chunk 0 and chunk 1 of the stage)
                if ((q == 0) != LATE) claim_b(IC<2 * TN + SC * RAW_IT>{}, bq[q & 1]);
                else claim_b(IC<2 * TN>{}, bq[q & 1]);
                __builtin_amdgcn_sched_barrier(0);
                const unsigned char* raw = stage + q * CHUNK_BYTES;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const unsigned char* rm = raw + m * (8 * PLANE_ROW * 16);
                    f32x4 e[3];
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const f32x4 d0 = *reinterpret_cast<const f32x4*>(rm + a_off[0][j]);
                        const f32x4 d1 = *reinterpret_cast<const f32x4*>(rm + a_off[1][j]);
                        e[j] = XI == 1 ? d0 + d1 : XI == 2 ? d1 - d0 : d0 - d1;
                    }
                    const f32x4 va = PB ? e[1] - e[0] : e[0] - e[2];
                    const f32x4 vb = PB ? e[0] - e[2] : e[1] + e[2];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int n = 0; n < TN; ++n) {
                            acc[0][m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[k], bq[q & 1][0][n][k], acc[0][m][n], 0, 0, 0);
                            acc[1][m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(vb[k], bq[q & 1][1][n][k], acc[1][m][n], 0, 0, 0);
                        }
                }
                if ((q == 0 && LATE) || (q == SC - 1 && !LATE)) {
                    // the registers hold stage st+1: park it in the other buffer (its readers passed the previous barrier)
                    store_raw((st & 1) ^ 1);
#pragma unroll
                    for (int q2 = 0; q2 < SC; ++q2) load_raw(rraw[q2]);
                }
            }
            __syncthreads();
        }
    };
    switch (wave) {          // wave-uniform
        case 0: k_loop(IC<0>{}, IC<0>{}); break;
        case 1: k_loop(IC<0>{}, IC<1>{}); break;
        case 2: k_loop(IC<1>{}, IC<0>{}); break;
        case 3: k_loop(IC<1>{}, IC<1>{}); break;
        case 4: k_loop(IC<2>{}, IC<0>{}); break;
        case 5: k_loop(IC<2>{}, IC<1>{}); break;
        case 6: k_loop(IC<3>{}, IC<0>{}); break;
        default: k_loop(IC<3>{}, IC<1>{}); break;
    }
    // The weight loads the last trip issued for the stage PAST the end (out of range: zeros) are still in flight.  For hipcc the
    // destination of an asm load is written when the statement ends, so without a later use it may hand those registers to the
    // epilogue's address arithmetic and schedule that above any wait: the late zeros then land on top of the addresses whenever
    // memory is slow (what conv_wino_x3w_kernel did beside a second stream).  Wait, then name the registers --
    // they stay allocated until the data has landed; build.verify_exit_reuse() re-derives it from the disassembly.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (X3) {
#pragma unroll
        for (int b_ = 0; b_ < (2); ++b_)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int n = 0; n < (X3 ? TN : 1); ++n)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) E2_CLAIM_AFTER_LOOP(bw[b_][a][n][pl]);
    } else {
#pragma unroll
        for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int n = 0; n < TN; ++n) E2_CLAIM_AFTER_LOOP(bq[b_][a][n]);
    }

    // ---- epilogue: gather the 16 positions in LDS, inverse transform, bias, residual, activation, store
    float* E = reinterpret_cast<float*>(smem);
    const int HW = p.H * p.W;
#pragma unroll
    for (int nh = 0; nh < TN; ++nh) {
        if (nh) __syncthreads();
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tile = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    E[((2 * wave + a) * TILES + tile) * 32 + i] = acc[a][m][nh][r];
                }
        __syncthreads();
        // one (tile, 4 consecutive couts) item per thread: 16 ds_read_b128, A^T M A, 4 stores of 16 bytes
        const int cq = tid & 7, tile = tid >> 3;
        const int n = n0 + nh * 32 + cq * 4;
        const int oy = by * (8 * MT) + 2 * (tile >> 3), ox = bx * 16 + 2 * (tile & 7);
        // H and W are even: a tile is inside or outside the image as a whole
        if (tile < TILES && n < p.Cout_g && oy < p.H && ox < p.W) {
            f32x4 mm[16];
#pragma unroll
            for (int a = 0; a < 16; ++a) mm[a] = *reinterpret_cast<const f32x4*>(E + (a * TILES + tile) * 32 + cq * 4);
            // A^T = [1 1 1 0; 0 1 -1 -1]
            f32x4 t0[4], t1[4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                t0[nu] = mm[0 * 4 + nu] + mm[1 * 4 + nu] + mm[2 * 4 + nu];
                t1[nu] = mm[1 * 4 + nu] - mm[2 * 4 + nu] - mm[3 * 4 + nu];
            }
            f32x4 y[4];
            y[0] = t0[0] + t0[1] + t0[2];
            y[1] = t0[1] - t0[2] - t0[3];
            y[2] = t1[0] + t1[1] + t1[2];
            y[3] = t1[1] - t1[2] - t1[3];
            const int co = g * p.Cout_g + n;
            const long long pix0 = (long long)img * HW + (long long)oy * p.W + ox;
            const int pstep[4] = {0, 1, p.W, p.W + 1};
            const bool full = n + 3 < p.Cout_g;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
#pragma unroll
                for (int c = 0; c < 4; ++c) bv[c] = (full || n + c < p.Cout_g) ? p.bias[co + c] : 0.f;
            }
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const long long pix = pix0 + pstep[px];
                f32x4 v = y[px] + bv;
                if (p.act == E2FGVI_ACT_DCNPOST) {
                    const f32x4 fl = *reinterpret_cast<const f32x4*>(p.res + pix * 4);
                    const float flv[4] = {fl[0], fl[1], fl[2], fl[3]};
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = wino_dcn_post(v[c], co + c, p.Cout, flv, p.slope);
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
                    for (int c = 0; c < 4; ++c) v[c] = apply_act(v[c], p.act, p.slope);
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

#if E2_WINO_X3

// ---- split-operand Winograd, FOUR transform positions per wave (round 3, late).  The eight-wave kernel above is bound by the
// LDS reads of its fragment phase (12 ds_read_b128 per lane and stage: every patch element is read by six of the eight waves)
// and runs one workgroup per CU (156+ registers), so nothing overlaps its weight waits, parks and barriers.  Here a wave owns
// one patch-row pair (xi = wave) and all four column combinations: 16 reads (2 rows x 4 columns x 2 channel quads) feed four
// positions instead of 12 feeding two -- a third less LDS traffic per MFMA -- and a workgroup is four waves (256 threads) of
// one 8x16-pixel block x BN couts, two of which fit a CU with independent barriers.  Weights double-buffered in registers,
// issued a stage ahead, as in the eight-wave kernel (explicit vmcnt, re-counted by build.verify_wino_waits).
template <int BN>
__global__ __launch_bounds__(256, 2) void conv_wino_x3p4_kernel(const WinoParams p) {
    constexpr int NT = 256, MT = 1, SC = 2;
    constexpr int TN = BN / 32;
    constexpr int RAW_H = 8 * MT + 2;
    constexpr int PLANE_RAW = RAW_H * PLANE_ROW * 16;
    constexpr int PLANE_BYTES = PLANE_RAW + ((16 - PLANE_RAW % 128) + 128) % 128;
    constexpr int CHUNK_BYTES = 4 * PLANE_BYTES;
    constexpr int STAGE_BYTES = SC * CHUNK_BYTES;
    constexpr int TILES = 32 * MT;
    constexpr int EPI_BYTES = 16 * TILES * 32 * 4;
    constexpr int SMEM = (2 * STAGE_BYTES > EPI_BYTES) ? 2 * STAGE_BYTES : EPI_BYTES;
    constexpr int RAW_ITEMS = RAW_H * RAW_W * 2;
    constexpr int RAW_IT = (RAW_ITEMS + NT - 1) / NT;

    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];

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
    const int y0 = by * (8 * MT) - 1, x0 = bx * 16 - 1;

    unsigned raw_off[RAW_IT];
    int raw_dst[RAW_IT];
#pragma unroll
    for (int it = 0; it < RAW_IT; ++it) {
        const int item = tid + it * NT;
        const int kq = item & 1, px = item >> 1;
        const int py = px / RAW_W, pxx = px - py * RAW_W;
        const int gy = y0 + py, gx = x0 + pxx;
        const bool have = item < RAW_ITEMS;
        const bool in = have && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
        raw_off[it] = in ? (unsigned)((img * p.H + gy) * p.W + gx) : OOB;
        raw_dst[it] = have ? (kq * 2 + (pxx & 1)) * PLANE_BYTES + (py * PLANE_ROW + (pxx >> 1)) * 16 : -1;
    }
    const unsigned raw_kq16 = (unsigned)(tid & 1) * 16u;

    int s = 0, c0 = 0;
    const float* cur_src = p.src[0];
    unsigned cur_bytes = p.src_bytes[0];
    unsigned cur_ld4 = (unsigned)p.ld[0] * 4u;
    unsigned cur_chan = (unsigned)(p.coff[0] + g * p.cpg[0]) * 4u;
    int cur_cpg = p.cpg[0];
    int raw_left = p.nchunks;       // chunks of the walk still inside the layer: past the end the prefetch fetches out of range
    f32x4 rraw[SC][RAW_IT];
    auto load_raw = [&](f32x4 (&q)[RAW_IT]) {
        const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(cur_src, cur_bytes);
        const unsigned chan = cur_chan + (unsigned)c0 * 4u + raw_kq16;
        const bool cvalid = raw_left-- > 0 && c0 + (int)(raw_kq16 >> 2) < cur_cpg;
#pragma unroll
        for (int it = 0; it < RAW_IT; ++it) {
            unsigned off = (cvalid && raw_off[it] != OOB) ? raw_off[it] * cur_ld4 + chan : OOB;
            asm volatile("" : "+v"(off));
            q[it] = buf_load4(arsrc, off);
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

    // this wave's patch rows: B^T rows  0: d0 - d2   1: d1 + d2   2: -d1 + d2   3: d1 - d3
    const int xi = wave;
    const int ra0 = (xi == 0) ? 0 : 1, ra1 = (xi == 3) ? 3 : 2;
    const int i = lane & 31, h = lane >> 5;
    const int ty = i >> 3, tx = i & 7;
    int a_off[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int prow = 2 * ty + (r ? ra1 : ra0);
            a_off[r][j] = h * CHUNK_BYTES + (j & 1) * PLANE_BYTES + (prow * PLANE_ROW + tx + (j >> 1)) * 16;
        }
    const i32x4 wrsrc = rsrc_words(reinterpret_cast<const char*>(p.w) + (long long)g * p.wgroup_bytes, p.wgroup_bytes);
    const unsigned u_step = 96u * (unsigned)p.Npad * 16u;
    const unsigned u_plane = 2u * (unsigned)p.Npad * 16u;
    unsigned u_off[4][TN];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int col = n0 + n * 32 + i;
            u_off[a][n] = col < p.Npad ? (unsigned)((((4 * wave + a) * 6 + h) * p.Npad + col) * 16) : 0x80000000u;
        }
    f32x4 bw[2][4][TN][3];
    auto load_b3 = [&](int st, f32x4 (&q)[4][TN][3]) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    buf_load4_pinned(q[a][n][pl], wrsrc, u_off[a][n] + (unsigned)pl * u_plane + (unsigned)st * u_step);
    };
    f32x16 acc[4][TN];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][n][r] = 0.f;

    const int nstages = (p.nchunks + SC - 1) / SC;
    // prologue: stage 0 into LDS buffer 0, stage 0's weights, stage 1 into the staging registers (the order the first trip's
    // explicit wait counts on)
#pragma unroll
    for (int q = 0; q < SC; ++q) load_raw(rraw[q]);
    load_b3(0, bw[0]);
    store_raw(0);
#pragma unroll
    for (int q = 0; q < SC; ++q) load_raw(rraw[q]);
    __syncthreads();

    auto k_loop = [&](auto XI_) __attribute__((always_inline)) {
        constexpr int XI = decltype(XI_)::value;
        auto trip = [&](auto CUR_, int st) __attribute__((always_inline)) {
            constexpr int CUR = decltype(CUR_)::value;
            const unsigned char* stage = smem + CUR * STAGE_BYTES;
            load_b3(st + 1, bw[CUR ^ 1]);
            // this stage's weights were issued a stage ago; since then: the patch prefetch and the next stage's weights
            wait_vmcnt<12 * TN + SC * RAW_IT>();
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(bw[CUR][a][n][pl]));
            __builtin_amdgcn_sched_barrier(0);
            f32x4 e[2][4];
#pragma unroll
            for (int kq = 0; kq < 2; ++kq)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f32x4 d0 = *reinterpret_cast<const f32x4*>(stage + kq * (2 * PLANE_BYTES) + a_off[0][j]);
                    const f32x4 d1 = *reinterpret_cast<const f32x4*>(stage + kq * (2 * PLANE_BYTES) + a_off[1][j]);
                    e[kq][j] = XI == 1 ? d0 + d1 : XI == 2 ? d1 - d0 : d0 - d1;
                }
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                bf16x8 ah, am, al;
                {
                    const f32x4 v0 = nu == 0 ? e[0][0] - e[0][2] : nu == 1 ? e[0][1] + e[0][2] : nu == 2 ? e[0][2] - e[0][1] : e[0][1] - e[0][3];
                    const f32x4 v1 = nu == 0 ? e[1][0] - e[1][2] : nu == 1 ? e[1][1] + e[1][2] : nu == 2 ? e[1][2] - e[1][1] : e[1][1] - e[1][3];
                    wino_split8(v0, v1, ah, am, al);
                }
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    const bf16x8 uh = __builtin_bit_cast(bf16x8, bw[CUR][nu][n][0]), um = __builtin_bit_cast(bf16x8, bw[CUR][nu][n][1]),
                                 ul = __builtin_bit_cast(bf16x8, bw[CUR][nu][n][2]);
                    acc[nu][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, uh, acc[nu][n], 0, 0, 0);
                    acc[nu][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, ul, acc[nu][n], 0, 0, 0);
                    acc[nu][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, um, acc[nu][n], 0, 0, 0);
                    acc[nu][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, uh, acc[nu][n], 0, 0, 0);
                    acc[nu][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, um, acc[nu][n], 0, 0, 0);
                    acc[nu][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, uh, acc[nu][n], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            // park stage st + 1 (the compiler waits for its loads -- and, not knowing the asm loads, for the weights issued above
            // as well), fetch stage st + 2
            store_raw(CUR ^ 1);
#pragma unroll
            for (int q2 = 0; q2 < SC; ++q2) load_raw(rraw[q2]);
            __syncthreads();
        };
        for (int st = 0; st < nstages; st += 2) {
            trip(IC<0>{}, st);
            trip(IC<1>{}, st + 1);
        }
    };
    switch (wave) {          // wave-uniform
        case 0: k_loop(IC<0>{}); break;
        case 1: k_loop(IC<1>{}); break;
        case 2: k_loop(IC<2>{}); break;
        default: k_loop(IC<3>{}); break;
    }
    // the weights of the stage past the end are still in flight: wait, then name their registers so that they stay allocated until
    // the data has landed (see conv_wino_kernel)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_)
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) E2_CLAIM_AFTER_LOOP(bw[b_][a][n][pl]);

    // ---- epilogue: gather the 16 positions in LDS, inverse transform, bias, residual, activation, store
    float* E = reinterpret_cast<float*>(smem);
    const int HW = p.H * p.W;
#pragma unroll
    for (int nh = 0; nh < TN; ++nh) {
        if (nh) __syncthreads();
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int tile = (r & 3) + 8 * (r >> 2) + 4 * h;
                E[((4 * wave + a) * TILES + tile) * 32 + i] = acc[a][nh][r];
            }
        __syncthreads();
        // one (tile, 4 consecutive couts) item per thread: 16 ds_read_b128, A^T M A, 4 stores of 16 bytes
        const int cq = tid & 7, tile = tid >> 3;
        const int n = n0 + nh * 32 + cq * 4;
        const int oy = by * (8 * MT) + 2 * (tile >> 3), ox = bx * 16 + 2 * (tile & 7);
        // H and W are even: a tile is inside or outside the image as a whole
        if (tile < TILES && n < p.Cout_g && oy < p.H && ox < p.W) {
            f32x4 mm[16];
#pragma unroll
            for (int a = 0; a < 16; ++a) mm[a] = *reinterpret_cast<const f32x4*>(E + (a * TILES + tile) * 32 + cq * 4);
            // A^T = [1 1 1 0; 0 1 -1 -1]
            f32x4 t0[4], t1[4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                t0[nu] = mm[0 * 4 + nu] + mm[1 * 4 + nu] + mm[2 * 4 + nu];
                t1[nu] = mm[1 * 4 + nu] - mm[2 * 4 + nu] - mm[3 * 4 + nu];
            }
            f32x4 y[4];
            y[0] = t0[0] + t0[1] + t0[2];
            y[1] = t0[1] - t0[2] - t0[3];
            y[2] = t1[0] + t1[1] + t1[2];
            y[3] = t1[1] - t1[2] - t1[3];
            const int co = g * p.Cout_g + n;
            const long long pix0 = (long long)img * HW + (long long)oy * p.W + ox;
            const int pstep[4] = {0, 1, p.W, p.W + 1};
            const bool full = n + 3 < p.Cout_g;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
#pragma unroll
                for (int c = 0; c < 4; ++c) bv[c] = (full || n + c < p.Cout_g) ? p.bias[co + c] : 0.f;
            }
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const long long pix = pix0 + pstep[px];
                f32x4 v = y[px] + bv;
                if (p.act == E2FGVI_ACT_DCNPOST) {
                    const f32x4 fl = *reinterpret_cast<const f32x4*>(p.res + pix * 4);
                    const float flv[4] = {fl[0], fl[1], fl[2], fl[3]};
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = wino_dcn_post(v[c], co + c, p.Cout, flv, p.slope);
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
                    for (int c = 0; c < 4; ++c) v[c] = apply_act(v[c], p.act, p.slope);
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
#endif

#if E2_WINO_X3

// x = hi + mid + lo for four fp32 values (the half of wino_split8 that belongs to one channel quad): two packed bf16 pairs per plane
__device__ __forceinline__ void wino_split4(const f32x4& v, unsigned (&H)[2], unsigned (&M)[2], unsigned (&L)[2]) {
    // (plain scalars first: __builtin_bit_cast / element access through a vector reference)
    float x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) x[j] = v[j];
#pragma unroll
    for (int j = 0; j < 2; ++j) e2_split2(x[2 * j], x[2 * j + 1], H[j], M[j], L[j]);
}

// ---- split-operand Winograd, WIDE tile (round 4): one eight-wave workgroup = a 16x16-pixel block (64 Winograd tiles, two MFMA
// row-tiles) x 64 output channels -- the largest block whose 16 x 64 x 64 fp32 accumulators (128 registers per lane) fit two
// waves per SIMD.  Against the 8x16-pixel x 32-cout shapes above, every transform + split of a patch feeds four times the MFMAs
// (5 VALU per MFMA instead of 10-17), every 16-byte weight fragment feeds 12 MFMAs instead of 6, and a stage moves 119 KB
// (21 KB of patch + 98 KB of weights) per 384 MFMAs instead of 60 KB per 96: 39 B/clk per CU at the matrix rate, under what
// the L2 delivers (~56 B/clk per CU).  What made the shape impossible in round 3 was the DOUBLE-buffered weight registers
// (2 x 48 + 128 accumulators + the transform's working set): here the weights are SINGLE-buffered and reloaded in place --
// the six MFMAs of a (position, column tile) group of the stage's second row-tile are the last readers of the group's three
// plane registers, and the pinned loads of the next stage's planes into the same registers are issued right behind them (an
// in-flight MFMA has read its B operand long before a load can return).  Loads return in order, so the first row-tile of the
// next stage claims them group by group with vmcnt(9 / 6 / 3 / 0): every load has the rest of its stage's MFMAs, the stage
// barrier and the next stage's first fragment phase to land.  Wave w owns the transform positions 2w, 2w + 1 as in the
// eight-wave kernel; fragments are built one row-tile at a time, one channel quad at a time (12 fragment registers per
// position), the patch of stage st + 1 is parked between the two row-tiles' MFMA blocks.
template <int BN>
__global__ __launch_bounds__(512, 2) void conv_wino_x3w_kernel(const WinoParams p) {
    constexpr int MT = 2, SC = 2;
    constexpr int TN = BN / 32;
    constexpr int RAW_H = 8 * MT + 2;
    constexpr int PLANE_RAW = RAW_H * PLANE_ROW * 16;
    constexpr int PLANE_BYTES = PLANE_RAW + ((16 - PLANE_RAW % 128) + 128) % 128;
    constexpr int CHUNK_USED = 4 * PLANE_BYTES;
    constexpr int CHUNK_BYTES = (CHUNK_USED + 1023) / 1024 * 1024;       // whole 1-KiB LDS-DMA pieces
    constexpr int STAGE_BYTES = SC * CHUNK_BYTES;
    constexpr int NSTAGE = 3;
    constexpr int TILES = 32 * MT;
    constexpr int EPI_BYTES = 16 * TILES * 32 * 4;
    constexpr int SMEM = (NSTAGE * STAGE_BYTES > EPI_BYTES) ? NSTAGE * STAGE_BYTES : EPI_BYTES;
    constexpr int PIECES = CHUNK_BYTES / 1024;                            // wave w issues the pieces w, w + 8 of a chunk
    constexpr int NPMAX = (PIECES + 7) / 8;

    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];

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
    const int y0 = by * (8 * MT) - 1, x0 = bx * 16 - 1;

    // The raw patch goes global -> LDS by LDS-DMA (no staging registers: the 128 accumulator + 48 weight registers leave no room
    // for them): each lane fetches the 16-byte unit that belongs at its LDS position of piece j of its wave (unit U of the chunk
    // area = plane [kq][column parity], patch row, column pair; padding units and pixels outside the image fetch out of range =
    // zeros).  dma_pix: the source pixel, bit 31 set = nothing to fetch; bit 30 = the channel quad kq of the unit's plane.
    unsigned dma_pix[NPMAX];
    {
        constexpr int UP = PLANE_BYTES / 16;
#pragma unroll
        for (int j = 0; j < NPMAX; ++j) {
            const int U = (wave + 8 * j) * 64 + lane;
            const int plane = U / UP, r2 = U - plane * UP;
            const int py = r2 / PLANE_ROW, c = r2 - py * PLANE_ROW;
            const int pxx = 2 * c + (plane & 1);
            const int gy = y0 + py, gx = x0 + pxx;
            const bool in = plane < 4 && py < RAW_H && c < RAW_W / 2 && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            dma_pix[j] = in ? ((unsigned)((img * p.H + gy) * p.W + gx) | ((unsigned)(plane >> 1) << 30)) : OOB;
        }
    }
    const unsigned smem_lds = (unsigned)(unsigned long long)(wino_lds_void*)smem;
    // The source walk of the pieces (chunks are fetched strictly in order: 0, 1, 2, ...): the parameters of the source being
    // walked live in scalar registers and are re-read from the kernel arguments only when the walk crosses into the next source
    // (past the last chunk the pieces fetch out of range, and so do those stages' weights: zeros).
    int w_s = 0, w_c0 = 0;
    const float* w_src = p.src[0];
    unsigned w_bytes = p.src_bytes[0], w_ld4 = (unsigned)p.ld[0] * 4u, w_chan = (unsigned)(p.coff[0] + g * p.cpg[0]) * 4u;
    int w_cpg = p.cpg[0];
    int w_left = p.nchunks;         // chunks of the walk still inside the layer: the pieces of the two stages past the end fetch out of range
    // the pieces of wave WV of the NEXT chunk of the walk -> LDS chunk area lds_chunk; exactly NPW vector-memory instructions
    auto dma_chunk = [&](auto W_, auto NPW_, unsigned lds_chunk) __attribute__((always_inline)) {
        constexpr int WV = decltype(W_)::value, NPW = decltype(NPW_)::value;
        const i32x4 rs = rsrc_words(w_src, w_bytes);
        const unsigned chan = w_chan + (unsigned)w_c0 * 4u;
        const bool half = w_c0 + 4 >= w_cpg;         // a source may end in the middle of a chunk: its kq = 1 units are zeros
        const bool live = w_left-- > 0;              // past the end: zeros without a memory access (before: the first source re-read)
#pragma unroll
        for (int j = 0; j < NPW; ++j) {
            const unsigned pix = dma_pix[j] & 0x3FFFFFFFu, kq = (dma_pix[j] >> 30) & 1u;
            unsigned off = (live && (int)dma_pix[j] >= 0 && !(half && kq)) ? pix * w_ld4 + chan + kq * 16u : OOB;
            asm volatile("" : "+v"(off));
            dma_piece(rs, __builtin_amdgcn_readfirstlane(lds_chunk + (unsigned)((WV + 8 * j) * 1024)), off);
        }
        w_c0 += 8;
        if (w_c0 >= w_cpg) {
            w_c0 = 0;
            ++w_s;
            if (w_s == p.nsrc) w_s = 0;
            if (p.nsrc > 1) {
                // static indices, one arm per source: a dynamic index into the parameter block makes hipcc copy the block to scratch
                // memory (and walk it with vector loads); each arm is a handful of scalar loads from the kernel arguments
#define E2_WALK_ARM(S) { w_src = p.src[S]; w_bytes = p.src_bytes[S]; w_ld4 = (unsigned)p.ld[S] * 4u; \
                         w_chan = (unsigned)(p.coff[S] + g * p.cpg[S]) * 4u; w_cpg = p.cpg[S]; }
                if (w_s == 0) E2_WALK_ARM(0) else if (w_s == 1) E2_WALK_ARM(1) else if (w_s == 2) E2_WALK_ARM(2) else E2_WALK_ARM(3)
#undef E2_WALK_ARM
            }
        }
    };

    const int i = lane & 31, h = lane >> 5;
    const int ty = i >> 3, tx = i & 7;
    // LDS byte offset of the lane's tile inside a stage: chunk h of the stage (8 channels), patch row 2 ty, column pair tx; the
    // patch row / column / channel quad / row-tile of a read are compile-time offsets on top
    const int a_lane = h * CHUNK_BYTES + (2 * ty * PLANE_ROW + tx) * 16;

    // weights: [stage][a = 16][plane = 3][h = 2][Npad] x 16 bytes.  ONE lane-dependent byte offset (half h, column i of the
    // workgroup's first column tile, advanced by a stage per trip: past the last stage it is out of the group's range and the loads
    // return zeros); position, plane and column tile are wave-uniform and go into the instruction's scalar offset (which the
    // buffer range check does not see: column tiles past Npad are redirected to tile 0 below)
    const i32x4 wrsrc = rsrc_words(reinterpret_cast<const char*>(p.w) + (long long)g * p.wgroup_bytes, p.wgroup_bytes);
    const unsigned u_step = 96u * (unsigned)p.Npad * 16u;
    const unsigned u_plane = 2u * (unsigned)p.Npad * 16u;
    unsigned u_lane = (unsigned)((h * p.Npad + n0 + i) * 16);
    auto load_plane = [&](f32x4& v, unsigned soff) __attribute__((always_inline)) {
        // s_nop 4: the scalar offset may be a spilled SGPR that the compiler has just restored with v_readlane (a VALU write of an
        // SGPR needs 5 wait states before a vector-memory instruction reads it, and the hazard recognizer does not look inside an
        // asm statement: without the nops, single plane loads used a stale offset in a few workgroups per launch)
        asm volatile("s_nop 4\n\tbuffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(v) : "v"(u_lane), "s"(wrsrc), "s"(soff) : "memory");
    };
    f32x4 bw[2][TN][3];
    f32x16 acc[2][MT][TN];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[a][m][n][r] = 0.f;

    const int nstages = (p.nchunks + SC - 1) / SC;

    auto k_loop = [&](auto XI_, auto PB_) __attribute__((always_inline)) {
        constexpr int XI = decltype(XI_)::value;
        constexpr bool PB = decltype(PB_)::value != 0;
        constexpr int R0 = (XI == 0) ? 0 : 1, R1 = (XI == 3) ? 3 : 2;       // the two patch rows of B^T row xi
        constexpr int CB = PB ? 1 : 0;                                      // first of the three patch columns
        constexpr int WV = 2 * XI + (PB ? 1 : 0);                           // this wave
        constexpr int NPW = (PIECES - WV + 7) / 8;                          // its LDS-DMA pieces per chunk
        const unsigned u_pos = (unsigned)(WV * 2 * 6) * (unsigned)p.Npad * 16u;     // scalar offset of position a = 0
        // ... of column tile n: a tile that lies entirely past Npad (Cout_g not a multiple of BN) re-reads tile 0's columns -- its
        // results are never stored, and the scalar offset must not lead outside the packed weights
        unsigned u_n[TN];
#pragma unroll
        for (int n = 0; n < TN; ++n) u_n[n] = (n0 + n * 32 < p.Npad) ? (unsigned)(n * 32 * 16) : 0u;
        // prologue: stages 0 and 1 of the patch, stage 0's planes; everything waited for
#pragma unroll
        for (int c4 = 0; c4 < 2 * SC; ++c4)
            dma_chunk(IC<WV>{}, IC<NPW>{}, smem_lds + (unsigned)((c4 / SC) * STAGE_BYTES + (c4 % SC) * CHUNK_BYTES));
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    load_plane(bw[a][n][pl], u_pos + (unsigned)(a * 3 + pl) * u_plane + u_n[n]);
        u_lane += u_step;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // fragments of one row-tile: positions a = 0, 1 x planes hi, mid, lo; built one channel quad at a time
        auto prep = [&](const unsigned char* rm, bf16x8 (&A)[6]) __attribute__((always_inline)) {
            unsigned P[6][4];
#pragma unroll
            for (int kq = 0; kq < 2; ++kq) {
                f32x4 e[3];
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const int col = CB + j;
                    const int off = kq * (2 * PLANE_BYTES) + (col & 1) * PLANE_BYTES + (col >> 1) * 16;
                    const f32x4 d0 = *reinterpret_cast<const f32x4*>(rm + off + R0 * PLANE_ROW * 16);
                    const f32x4 d1 = *reinterpret_cast<const f32x4*>(rm + off + R1 * PLANE_ROW * 16);
                    e[j] = XI == 1 ? d0 + d1 : XI == 2 ? d1 - d0 : d0 - d1;
                }
                const f32x4 va = PB ? e[1] - e[0] : e[0] - e[2];
                const f32x4 vb = PB ? e[0] - e[2] : e[1] + e[2];
                unsigned H[2], M[2], L[2];
                wino_split4(va, H, M, L);
                P[0][2 * kq] = H[0]; P[0][2 * kq + 1] = H[1]; P[1][2 * kq] = M[0]; P[1][2 * kq + 1] = M[1];
                P[2][2 * kq] = L[0]; P[2][2 * kq + 1] = L[1];
                wino_split4(vb, H, M, L);
                P[3][2 * kq] = H[0]; P[3][2 * kq + 1] = H[1]; P[4][2 * kq] = M[0]; P[4][2 * kq + 1] = M[1];
                P[5][2 * kq] = L[0]; P[5][2 * kq + 1] = L[1];
            }
#pragma unroll
            for (int f = 0; f < 6; ++f) {
                u32x4 t = {P[f][0], P[f][1], P[f][2], P[f][3]};
                A[f] = __builtin_bit_cast(bf16x8, t);
            }
        };
        // the 12 TN MFMAs of one row-tile; LAST: behind each position's block, the next stage's planes into the same registers
        auto mma = [&](auto M_, auto LAST_, const bf16x8 (&A)[6]) __attribute__((always_inline)) {
            constexpr int M = decltype(M_)::value;
            constexpr bool LAST = decltype(LAST_)::value != 0;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                if constexpr (!LAST) {
                    // this position's 3 TN plane loads were issued a stage ago, and after them the other position's (a = 0: 3 TN)
                    // (issued twice: the mark build.verify_wino_waits() looks for)
                    if (a == 0) asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt vmcnt(%0)" ::"n"(3 * TN) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                    for (int n = 0; n < TN; ++n)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl) asm volatile("" : "+v"(bw[a][n][pl]));
                }
                __builtin_amdgcn_sched_barrier(0);     // the positions' MFMA blocks stay behind their own waits
                const bf16x8 xh = A[3 * a], xm = A[3 * a + 1], xl = A[3 * a + 2];
                bf16x8 uh[TN], um[TN], ul[TN];
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    uh[n] = __builtin_bit_cast(bf16x8, bw[a][n][0]);
                    um[n] = __builtin_bit_cast(bf16x8, bw[a][n][1]);
                    ul[n] = __builtin_bit_cast(bf16x8, bw[a][n][2]);
                }
                // smallest terms first, the column tiles interleaved (independent accumulators)
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[a][M][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, uh[n], acc[a][M][n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[a][M][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, ul[n], acc[a][M][n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[a][M][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, um[n], acc[a][M][n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[a][M][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, uh[n], acc[a][M][n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[a][M][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, um[n], acc[a][M][n], 0, 0, 0);
#pragma unroll
                for (int n = 0; n < TN; ++n) acc[a][M][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, uh[n], acc[a][M][n], 0, 0, 0);
                if constexpr (LAST) {
                    __builtin_amdgcn_sched_barrier(0);
                    // ONE load per asm statement, each behind its own s_nop 4 (a restored scalar offset: load_plane).  The reloads of
                    // the LAST stage target the stage past the end (out of range: zeros) and are still in flight behind the loop:
                    // see the wait + register claims in front of the epilogue (the cause of round 4's wrong blocks beside a second
                    // stream).
#pragma unroll
                    for (int n = 0; n < TN; ++n)
#pragma unroll
                        for (int pl = 0; pl < 3; ++pl)
                            load_plane(bw[a][n][pl], u_pos + (unsigned)(a * 3 + pl) * u_plane + u_n[n]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        // One stage = P0 (fragments of row-tile 0) M0 (its MFMAs) P1 (fragments of row-tile 1 + this wave's LDS-DMA pieces of stage
        // st + 2) M1 (its MFMAs + the plane reloads), one barrier per stage.  Stage st lives in LDS slot st % 3; stage st + 2 goes to
        // slot (st + 2) % 3, whose readers passed the last barrier.
        // (Measured and dropped, profiles/r04_x3w_variants.txt: the phases of waves 4-7 shifted by one against their SIMD partners,
        //  s_setprio 3 around the MFMA blocks, both: all within +-3 %.)
        constexpr bool SKEW = false;
        int slot = 0;
        auto first_half = [&](bf16x8 (&A1)[6]) __attribute__((always_inline)) {
            const unsigned char* rm = smem + slot * STAGE_BYTES + a_lane;
            const unsigned ahead = smem_lds + (unsigned)((slot == 0 ? 2 : slot - 1) * STAGE_BYTES);
            {
                bf16x8 A[6];
                prep(rm, A);
                __builtin_amdgcn_sched_barrier(0);
                mma(IC<0>{}, IC<0>{}, A);
            }
            __builtin_amdgcn_sched_barrier(0);
            prep(rm + 8 * PLANE_ROW * 16, A1);
            __builtin_amdgcn_sched_barrier(0);
            // this wave's pieces of stage st + 2: in front of the plane loads, so that the next stage's first plane wait also
            // retires them (loads return in order); the barrier at the end of the next stage publishes them
#pragma unroll
            for (int q = 0; q < SC; ++q) dma_chunk(IC<WV>{}, IC<NPW>{}, ahead + (unsigned)(q * CHUNK_BYTES));
            __builtin_amdgcn_sched_barrier(0);
        };
        auto second_half = [&](const bf16x8 (&A1)[6]) __attribute__((always_inline)) {
            mma(IC<1>{}, IC<1>{}, A1);
            u_lane += u_step;
            slot = slot == 2 ? 0 : slot + 1;
        };
        auto stage_barrier = [&]() __attribute__((always_inline)) {
            __syncthreads();
        };
        if constexpr (SKEW) {
            bf16x8 A1[6];
            first_half(A1);
            stage_barrier();
            for (int st = 0; st < nstages; ++st) {
                second_half(A1);
                if (st + 1 < nstages) {           // (the last M1 runs into the barrier behind the loop: as many barriers as the other waves)
                    first_half(A1);
                    stage_barrier();
                }
            }
        } else {
            for (int st = 0; st < nstages; ++st) {
                bf16x8 A1[6];
                first_half(A1);
                second_half(A1);
                stage_barrier();
            }
        }
    };
    switch (wave) {          // wave-uniform
        case 0: k_loop(IC<0>{}, IC<0>{}); break;
        case 1: k_loop(IC<0>{}, IC<1>{}); break;
        case 2: k_loop(IC<1>{}, IC<0>{}); break;
        case 3: k_loop(IC<1>{}, IC<1>{}); break;
        case 4: k_loop(IC<2>{}, IC<0>{}); break;
        case 5: k_loop(IC<2>{}, IC<1>{}); break;
        case 6: k_loop(IC<3>{}, IC<0>{}); break;
        default: k_loop(IC<3>{}, IC<1>{}); break;
    }
    // The planes and the patch pieces of the stages past the end are still in flight: they target registers and LDS the epilogue
    // reuses.  The plane registers are named behind the wait: for hipcc an asm load's destination is written when the statement
    // ends, so without a later use it took bw's registers for the epilogue's address arithmetic and hoisted that above this
    // wait -- the planes of the stage past the end (out of range: zeros) then landed ON TOP of those addresses whenever the memory
    // system was slow enough to deliver them late: wrong output blocks beside a second stream (in every launch beside a device copy:
    // tools/c4_repro.py), never alone.  DESIGN.md C4.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) E2_CLAIM_AFTER_LOOP(bw[a][n][pl]);
    __syncthreads();

    // ---- epilogue: gather the 16 positions in LDS, inverse transform, bias, residual, activation, store
    float* E = reinterpret_cast<float*>(smem);
    const int HW = p.H * p.W;
#pragma unroll
    for (int nh = 0; nh < TN; ++nh) {
        if (nh) __syncthreads();
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int tile = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    E[((2 * wave + a) * TILES + tile) * 32 + i] = acc[a][m][nh][r];
                }
        __syncthreads();
        const int cq = tid & 7, tile = tid >> 3;
        const int n = n0 + nh * 32 + cq * 4;
        const int oy = by * (8 * MT) + 2 * (tile >> 3), ox = bx * 16 + 2 * (tile & 7);
        if (tile < TILES && n < p.Cout_g && oy < p.H && ox < p.W) {
            f32x4 mm[16];
#pragma unroll
            for (int a = 0; a < 16; ++a) mm[a] = *reinterpret_cast<const f32x4*>(E + (a * TILES + tile) * 32 + cq * 4);
            f32x4 t0[4], t1[4];
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) {
                t0[nu] = mm[0 * 4 + nu] + mm[1 * 4 + nu] + mm[2 * 4 + nu];
                t1[nu] = mm[1 * 4 + nu] - mm[2 * 4 + nu] - mm[3 * 4 + nu];
            }
            f32x4 y[4];
            y[0] = t0[0] + t0[1] + t0[2];
            y[1] = t0[1] - t0[2] - t0[3];
            y[2] = t1[0] + t1[1] + t1[2];
            y[3] = t1[1] - t1[2] - t1[3];
            const int co = g * p.Cout_g + n;
            const long long pix0 = (long long)img * HW + (long long)oy * p.W + ox;
            const int pstep[4] = {0, 1, p.W, p.W + 1};
            const bool full = n + 3 < p.Cout_g;
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (p.bias) {
#pragma unroll
                for (int c = 0; c < 4; ++c) bv[c] = (full || n + c < p.Cout_g) ? p.bias[co + c] : 0.f;
            }
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const long long pix = pix0 + pstep[px];
                f32x4 v = y[px] + bv;
                if (p.act == E2FGVI_ACT_DCNPOST) {
                    const f32x4 fl = *reinterpret_cast<const f32x4*>(p.res + pix * 4);
                    const float flv[4] = {fl[0], fl[1], fl[2], fl[3]};
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[c] = wino_dcn_post(v[c], co + c, p.Cout, flv, p.slope);
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
                    for (int c = 0; c < 4; ++c) v[c] = apply_act(v[c], p.act, p.slope);
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
#endif

struct WinoPack {
    int Cout, groups, nsrc;
    int cpg[E2FGVI_MAX_SRC];
    int Cout_g, Npad, Cin_g, nchunks;
    long long wgroup_elems, total;
};

bool wino_geometry(int Cout, int groups, int nsrc, const int32_t* cpg, WinoPack* q) {
    if (Cout <= 0 || groups <= 0 || Cout % groups || nsrc < 1 || nsrc > E2FGVI_MAX_SRC) return false;
    q->Cout = Cout; q->groups = groups; q->nsrc = nsrc;
    q->Cout_g = Cout / groups;
    q->Npad = round_up(q->Cout_g, 32);
    q->Cin_g = 0; q->nchunks = 0;
    for (int s = 0; s < E2FGVI_MAX_SRC; ++s) q->cpg[s] = 0;
    for (int s = 0; s < nsrc; ++s) {
        if (cpg[s] <= 0 || cpg[s] % 4) return false;
        q->cpg[s] = cpg[s];
        q->Cin_g += cpg[s];
        q->nchunks += (cpg[s] + 7) / 8;          // a source ends its last chunk half empty when cpg % 8 == 4
    }
    q->wgroup_elems = (long long)q->nchunks * 16 * 2 * q->Npad * 4;
    q->total = q->wgroup_elems * groups;
    return true;
}

__global__ void pack_wino_weight_kernel(const float* __restrict__ w, float* __restrict__ wp, const WinoPack p) {
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
    const int a = (int)(rem & 15);
    const int chunk = (int)(rem >> 4);
    // chunk -> (source, channel inside the group's slice of that source)
    int s = 0, prefix = 0, lc = chunk;                     // chunk -> (source, chunk inside the source)
    while (lc >= (p.cpg[s] + 7) / 8) { lc -= (p.cpg[s] + 7) / 8; prefix += p.cpg[s]; ++s; }
    const int ch = lc * 8 + kq * 4 + kk;
    float v = 0.f;
    if (n < p.Cout_g && ch < p.cpg[s]) {
        const float* f = w + ((long long)(g * p.Cout_g + n) * p.Cin_g + prefix + ch) * 9;
        // G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  U = G f G^T
        const int xi = a >> 2, nu = a & 3;
        float col[3];          // (G f)[xi][j]
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float f0 = f[0 * 3 + j], f1 = f[1 * 3 + j], f2 = f[2 * 3 + j];
            col[j] = xi == 0 ? f0 : xi == 1 ? 0.5f * (f0 + f1 + f2) : xi == 2 ? 0.5f * (f0 - f1 + f2) : f2;
        }
        v = nu == 0 ? col[0] : nu == 1 ? 0.5f * (col[0] + col[1] + col[2]) : nu == 2 ? 0.5f * (col[0] - col[1] + col[2]) : col[2];
    }
    wp[idx] = v;
}

// the transformed weight U[a] of (group g, output column n, input channel `ch` of source s) -- pack_wino_weight_kernel's formula
__device__ __forceinline__ float wino_u(const float* __restrict__ w, const WinoPack& p, int g, int n, int prefix, int ch, int a) {
    const float* f = w + ((long long)(g * p.Cout_g + n) * p.Cin_g + prefix + ch) * 9;
    const int xi = a >> 2, nu = a & 3;
    float col[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float f0 = f[0 * 3 + j], f1 = f[1 * 3 + j], f2 = f[2 * 3 + j];
        col[j] = xi == 0 ? f0 : xi == 1 ? 0.5f * (f0 + f1 + f2) : xi == 2 ? 0.5f * (f0 - f1 + f2) : f2;
    }
    return nu == 0 ? col[0] : nu == 1 ? 0.5f * (col[0] + col[1] + col[2]) : nu == 2 ? 0.5f * (col[0] - col[1] + col[2]) : col[2];
}

// X3 packing: [group][stage][a = 16][plane][h][Npad][8 bf16], stage = chunks 2 stage + h, element j = channel j of the chunk;
// one thread per fp32 value, which it splits into its three bf16 pieces (their sum is the value, bit for bit)
__global__ void pack_wino_weight_x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, const WinoPack p, int nstages) {
    const long long per_group = (long long)nstages * 16 * 2 * p.Npad * 8;          // values (not planes) per group
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= per_group * p.groups) return;
    const int g = (int)(idx / per_group);
    long long rem = idx - (long long)g * per_group;
    const int j = (int)(rem & 7);
    rem >>= 3;
    const int n = (int)(rem % p.Npad);
    rem /= p.Npad;
    const int h = (int)(rem & 1);
    rem >>= 1;
    const int a = (int)(rem & 15);
    const int stage = (int)(rem >> 4);
    const int chunk = 2 * stage + h;
    float v = 0.f;
    if (chunk < p.nchunks && n < p.Cout_g) {
        int s = 0, prefix = 0, lc = chunk;
        while (lc >= (p.cpg[s] + 7) / 8) { lc -= (p.cpg[s] + 7) / 8; prefix += p.cpg[s]; ++s; }
        const int ch = lc * 8 + j;
        if (ch < p.cpg[s]) v = wino_u(w, p, g, n, prefix, ch, a);
    }
    const unsigned xb = __builtin_bit_cast(unsigned, v);
    const float r = v - __builtin_bit_cast(float, xb & 0xFFFF0000u);
    const unsigned rb = __builtin_bit_cast(unsigned, r);
    const float r2 = r - __builtin_bit_cast(float, rb & 0xFFFF0000u);
    const long long plane = 2LL * p.Npad * 8;
    unsigned short* o = wp + (long long)g * per_group * 3 + ((long long)(stage * 16 + a) * 3) * plane + ((long long)h * p.Npad + n) * 8 + j;
    o[0] = (unsigned short)(xb >> 16);
    o[plane] = (unsigned short)(rb >> 16);
    o[2 * plane] = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
}

}  // namespace

#if E2_WINO_X3
/* split-bf16 Winograd (ABI version 7): weights as three bf16 planes of the transformed fp32 weights; returns bf16 ELEMENTS */
extern "C" int64_t e2fgvi_packed_winograd_weight_x3_size(int32_t Cout, int32_t groups, int32_t nsrc, const int32_t* src_cpg) {
    WinoPack q;
    if (!src_cpg || !wino_geometry(Cout, groups, nsrc, src_cpg, &q)) {
        e2fgvi_set_error("packed_winograd_weight_x3_size: bad geometry (channels per source must be multiples of 4)");
        return E2FGVI_EINVAL;
    }
    return (long long)((q.nchunks + 1) / 2) * 16 * 3 * 2 * q.Npad * 8 * groups;
}

extern "C" int e2fgvi_pack_winograd_weight_x3(const float* w, void* wpacked, int32_t Cout, int32_t groups, int32_t nsrc,
                                              const int32_t* src_cpg, void* stream) {
    WinoPack q;
    E2_REQUIRE(w && wpacked && src_cpg, E2FGVI_EINVAL, "pack_winograd_weight_x3: null pointer");
    E2_REQUIRE(wino_geometry(Cout, groups, nsrc, src_cpg, &q), E2FGVI_EINVAL, "pack_winograd_weight_x3: bad geometry");
    const int nstages = (q.nchunks + 1) / 2;
    const long long values = (long long)nstages * 16 * 2 * q.Npad * 8 * groups;
    hipLaunchKernelGGL(pack_wino_weight_x3_kernel, dim3((unsigned)cdiv64(values, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       (unsigned short*)wpacked, q, nstages);
    E2_LAUNCH_CHECK("pack_winograd_weight_x3");
    return 0;
}
#else
extern "C" int64_t e2fgvi_packed_winograd_weight_size(int32_t Cout, int32_t groups, int32_t nsrc, const int32_t* src_cpg) {
    WinoPack q;
    if (!src_cpg || !wino_geometry(Cout, groups, nsrc, src_cpg, &q)) {
        e2fgvi_set_error("packed_winograd_weight_size: bad geometry (channels per source must be multiples of 8)");
        return E2FGVI_EINVAL;
    }
    return q.total;
}

extern "C" int e2fgvi_pack_winograd_weight(const float* w, float* wpacked, int32_t Cout, int32_t groups, int32_t nsrc,
                                           const int32_t* src_cpg, void* stream) {
    WinoPack q;
    E2_REQUIRE(w && wpacked && src_cpg, E2FGVI_EINVAL, "pack_winograd_weight: null pointer");
    E2_REQUIRE(wino_geometry(Cout, groups, nsrc, src_cpg, &q), E2FGVI_EINVAL, "pack_winograd_weight: bad geometry");
    hipLaunchKernelGGL(pack_wino_weight_kernel, dim3((unsigned)cdiv64(q.total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                       wpacked, q);
    E2_LAUNCH_CHECK("pack_winograd_weight");
    return 0;
}
#endif

template <int MT, int BN, int SC, bool X3 = false>
static int launch_wino(WinoParams& p, int groups, hipStream_t st) {
    p.blocksY = cdiv(p.H, 8 * MT);
    p.blocksX = cdiv(p.W, 16);
    p.tilesN = cdiv(p.Cout_g, BN);
    const long long nblk = (long long)p.N * p.blocksY * p.blocksX * p.tilesN;
    E2_REQUIRE(nblk < 2147483647LL, E2FGVI_EUNSUP, "conv3x3_winograd: grid too large");
    p.nblk = (int)nblk;
    hipLaunchKernelGGL((conv_wino_kernel<MT, BN, SC, X3>), dim3(p.nblk, groups, 1), dim3(512), 0, st, p);
    E2_LAUNCH_CHECK("conv3x3_winograd");
    return 0;
}

#if E2_WINO_X3
template <int BN>
static int launch_wino_p4(WinoParams& p, int groups, hipStream_t st) {
    p.blocksY = cdiv(p.H, 8);
    p.blocksX = cdiv(p.W, 16);
    p.tilesN = cdiv(p.Cout_g, BN);
    const long long nblk = (long long)p.N * p.blocksY * p.blocksX * p.tilesN;
    E2_REQUIRE(nblk < 2147483647LL, E2FGVI_EUNSUP, "conv3x3_winograd_x3: grid too large");
    p.nblk = (int)nblk;
    hipLaunchKernelGGL((conv_wino_x3p4_kernel<BN>), dim3(p.nblk, groups, 1), dim3(256), 0, st, p);
    E2_LAUNCH_CHECK("conv3x3_winograd_x3 (four positions per wave)");
    return 0;
}
#endif

#if E2_WINO_X3
template <int BN>
static int launch_wino_w(WinoParams& p, int groups, hipStream_t st) {
    p.blocksY = cdiv(p.H, 16);
    p.blocksX = cdiv(p.W, 16);
    p.tilesN = cdiv(p.Cout_g, BN);
    const long long nblk = (long long)p.N * p.blocksY * p.blocksX * p.tilesN;
    E2_REQUIRE(nblk < 2147483647LL, E2FGVI_EUNSUP, "conv3x3_winograd_x3: grid too large");
    p.nblk = (int)nblk;
    hipLaunchKernelGGL((conv_wino_x3w_kernel<BN>), dim3(p.nblk, groups, 1), dim3(512), 0, st, p);
    E2_LAUNCH_CHECK("conv3x3_winograd_x3 (wide tile)");
    return 0;
}
#endif

static int wino_run(const e2fgvi_conv_desc* d, void* stream, bool x3) {
    E2_REQUIRE(d, E2FGVI_EINVAL, "conv3x3_winograd: null descriptor");
    WinoPack q;
    E2_REQUIRE(wino_geometry(d->Cout, d->groups, d->nsrc, d->src_cpg, &q), E2FGVI_EINVAL,
               "conv3x3_winograd: bad geometry (channels per source must be multiples of 4)");
    E2_REQUIRE(d->KH == 3 && d->KW == 3 && d->stride == 1 && d->pad == 1, E2FGVI_EUNSUP,
               "conv3x3_winograd: only 3x3, stride 1, pad 1");
    E2_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->H % 2 == 0 && d->W % 2 == 0, E2FGVI_EUNSUP,
               "conv3x3_winograd: H and W must be even");
    E2_REQUIRE(d->Ho == d->H && d->Wo == d->W, E2FGVI_EINVAL, "conv3x3_winograd: Ho/Wo must equal H/W");
    E2_REQUIRE(d->wpacked && d->dst, E2FGVI_EINVAL, "conv3x3_winograd: null weight/dst");
    E2_REQUIRE(!d->dst_nchw, E2FGVI_EUNSUP, "conv3x3_winograd: NCHW output not supported");
    E2_REQUIRE(d->dst_coff >= 0 && d->dst_coff + d->Cout <= d->dst_ld, E2FGVI_EINVAL, "conv3x3_winograd: dst slice exceeds dst_ld");
    if (d->act == E2FGVI_ACT_DCNPOST)
        E2_REQUIRE(d->residual && d->Cout % 3 == 0 && d->groups == 1 && ((uintptr_t)d->residual & 15) == 0, E2FGVI_EINVAL,
                   "conv3x3_winograd: ACT_DCNPOST needs the [pixel][4] flows as residual, Cout %% 3 == 0, groups == 1");
    WinoParams p;
    for (int s = 0; s < E2FGVI_MAX_SRC; ++s) { p.src[s] = nullptr; p.ld[s] = 0; p.coff[s] = 0; p.cpg[s] = 0; p.src_bytes[s] = 0; }
    for (int s = 0; s < d->nsrc; ++s) {
        E2_REQUIRE(d->src[s], E2FGVI_EINVAL, "conv3x3_winograd: null source %d", s);
        E2_REQUIRE(d->src_ld[s] % 4 == 0 && d->src_coff[s] % 4 == 0 && ((uintptr_t)d->src[s] & 15) == 0, E2FGVI_EINVAL,
                   "conv3x3_winograd: source %d not 16-byte addressable", s);
        E2_REQUIRE(d->src_coff[s] + d->groups * d->src_cpg[s] <= d->src_ld[s], E2FGVI_EINVAL,
                   "conv3x3_winograd: source %d channel range exceeds its pixel stride", s);
        const long long bytes = (long long)d->N * d->H * d->W * d->src_ld[s] * 4;
        E2_REQUIRE(bytes < 4294967295LL, E2FGVI_EUNSUP, "conv3x3_winograd: source %d spans >= 4 GiB (split the batch)", s);
        p.src[s] = d->src[s]; p.ld[s] = d->src_ld[s]; p.coff[s] = d->src_coff[s]; p.cpg[s] = d->src_cpg[s];
        p.src_bytes[s] = (unsigned)bytes;
    }
    E2_REQUIRE(q.wgroup_elems * 4 < 0x70000000LL, E2FGVI_EUNSUP, "conv3x3_winograd: packed weight group >= 1.75 GiB");
    E2_REQUIRE(((uintptr_t)d->wpacked & 15) == 0, E2FGVI_EINVAL, "conv3x3_winograd: packed weight not 16-byte aligned");
    p.nsrc = d->nsrc;
    p.N = d->N; p.H = d->H; p.W = d->W;
    p.Cout = d->Cout; p.Cout_g = q.Cout_g; p.Npad = q.Npad;
    p.nchunks = q.nchunks;
    p.wgroup_elems = q.wgroup_elems; p.wgroup_bytes = (unsigned)(q.wgroup_elems * 4);
    if (x3) p.wgroup_bytes = (unsigned)((long long)((q.nchunks + 1) / 2) * 16 * 3 * 2 * q.Npad * 16);   // [stage][a][plane][h][Npad] x 16 bytes
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
    if (!tile) {
        // Measured on MI355X (tools/wino_bench.py, profiles/r01_wino_bench.txt): the 8x16-pixel x 32-cout shape (two
        // workgroups per CU: one's prologue / epilogue hides under the other's MFMAs) wins or ties everywhere except on
        // the layers with >= 256 output channels per group, where the 16x16 x 64 shape amortises the patch transform
        // over twice the columns and is 3-6 % faster.
        const long long big = (long long)d->N * cdiv(d->H, 16) * cdiv(d->W, 16) * cdiv(q.Cout_g, 64) * d->groups;
        tile = (q.Cout_g >= 256 && big >= 128) ? 64 : 132;
    }
#if E2_WINO_X3
    (void)x3;
    if (tile == 64) tile = 164;
    switch (tile) {
        // (no 16x16-pixel x 64-cout shape: 128 accumulator + 96 weight registers per lane do not fit beside the split)
        case 32: return launch_wino<2, 32, 2, true>(p, d->groups, st);
        case 164: return launch_wino<1, 64, 2, true>(p, d->groups, st);
        case 132: return launch_wino<1, 32, 2, true>(p, d->groups, st);
        // + 5000: four positions per wave, four-wave workgroups (two per CU)
        case 5132: return launch_wino_p4<32>(p, d->groups, st);
        // + 6000: 16x16-pixel blocks x 64 couts, single-buffered weights reloaded in place (round 4)
        case 6064: return launch_wino_w<64>(p, d->groups, st);
        default: break;
    }
    e2fgvi_set_error("conv3x3_winograd_x3: tile must be 0 (auto), 32 (16x16-pixel blocks) or 132, 164 (8x16-pixel blocks)");
    return E2FGVI_EINVAL;
#else
    switch (tile) {
        case 64: return launch_wino<2, 64, 2>(p, d->groups, st);
        case 32: return launch_wino<2, 32, 2>(p, d->groups, st);
        case 164: return launch_wino<1, 64, 2>(p, d->groups, st);
        case 132: return launch_wino<1, 32, 2>(p, d->groups, st);
        // (four chunks per LDS stage -- half the barriers -- measured 1-6 % slower on every layer, profiles/r02_wino_sc4.txt)
        default: break;
    }
    e2fgvi_set_error("conv3x3_winograd: tile must be 0 (auto), 32, 64 (16x16-pixel blocks) or 132, 164 (8x16-pixel blocks)");
    return E2FGVI_EINVAL;
#endif
}

#if E2_WINO_X3
/* e2fgvi_conv3x3_winograd on the bf16 matrix pipe: same descriptor, wpacked from e2fgvi_pack_winograd_weight_x3 */
extern "C" int e2fgvi_conv3x3_winograd_x3(const e2fgvi_conv_desc* d, void* stream) { return wino_run(d, stream, true); }
#else
extern "C" int e2fgvi_conv3x3_winograd(const e2fgvi_conv_desc* d, void* stream) { return wino_run(d, stream, false); }
#endif

