// Implicit-GEMM convolution / linear layer of the bf16 data path (gfx950): bf16 NHWC activations in HBM, bf16 packed
// weights, v_mfma_f32_32x32x16_bf16 with fp32 accumulation, fp32 epilogue, bf16 and / or fp32 stores.
//
// Both operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 16 bytes per lane, 1 KiB per wave
// instruction): no staging registers, no ds_write pass, out-of-range lanes (image halo, channel tails, rows past M,
// columns past Npad) are given an out-of-range buffer offset and land as zeros.  The DMA writes LDS linearly in lane
// order, so the bank-conflict-free image is produced on the SOURCE side: lane (row, slot) of the A tile fetches the
// 16-byte channel chunk  slot ^ ((row >> 1) & 7)  of its pixel -- the 8 lanes of a row still cover one contiguous
// 128-byte run of the pixel's channels, only permuted.
//   LDS A stage: [BM rows][8 chunks of 8 bf16]   chunk c of row r at 16-byte slot  r*8 + (c ^ ((r >> 1) & 7))
//   LDS B stage: [8 k-octets][BN][8 bf16]        (the packed weight layout, lane-linear)
//   MFMA operands (v_mfma_f32_32x32x16_bf16): lane l supplies 8 consecutive k (k-octet 2*kk + (l >> 5)) of row / column
//   (l & 31): one ds_read_b128 each; the XOR swizzle makes the 16-lane groups of a b128 read hit 16 distinct slots.
// K loop: one 64-deep step (64 input channels of one (tap, source)) per barrier, two LDS stages: the DMA of step s+1 is
// issued before the MFMAs of step s and drained (s_waitcnt vmcnt(0)) at the barrier that ends step s.
// Epilogue: every wave parks its accumulator tiles in LDS (the stages are free by then) and reads them back row-wise,
// 8 consecutive output channels per lane: bias, residual (fp32 or bf16), activation or the DCN offset / mask
// post-processing, then 16-byte bf16 stores and / or 2 x 16-byte fp32 stores -- whole 128 / 256-byte rows per wave.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

struct ConvXParams {
    const void* src[E2FGVI_MAX_SRC];
    int ld[E2FGVI_MAX_SRC];
    int coff[E2FGVI_MAX_SRC];
    int cpg[E2FGVI_MAX_SRC];
    unsigned src_bytes[E2FGVI_MAX_SRC];
    int nsrc;
    int N, H, W, Ho, Wo, KH, KW, stride, pad;
    int padx;                             // left padding (= pad unless the descriptor gives an explicit output grid)
    int osy, osx, opy, opx, oH, oW;       // osy > 0: output pixel (img, oy, ox) lives at (img, oy*osy + opy, ox*osx + opx) of [N,oH,oW]
    int res_bcast;                        // the residual is ONE [oH*oW] (or [Ho*Wo]) image added to every image of the batch
    int Cout, Cout_g, Npad;
    int M;
    int tilesM, tilesN;
    int nsteps;                           // KH * KW * (K-step blocks per tap); a K-step = 64 bf16 / 32 fp32 channels
    unsigned wgroup_bytes;
    long long wgroup_elems;
    const void* w;
    const float* bias;
    const void* res;
    int res_ld, res_coff, res_bf16;
    void* dst;
    int dst_ld, dst_coff, dst_bf16, dst_nchw;
    __bf16* dst2;
    int dst2_ld, dst2_coff;
    int split_from;                    // ABI 8: channels >= split_from go to dst2 as three exact bf16 planes (split_plane elements apart)
    long long split_plane;
    int act;
    float slope;
    int tp_cq;                            // tap-packed K-steps (narrow single-source layers): 16-byte chunks per tap (1..7), 0 = off
    unsigned tp_magic;                    //   ceil(2^32 / KW): tap -> kernel row by __umulhi
    unsigned tp_magic_cq;                 //   ceil(2^32 / cq): chunk of the stream -> tap (unused for cq = 1)
    int dbg_noload;                       // tools only (tile codes 21 / 26): skip every DMA after the first stage -> the MFMA + LDS ceiling
};

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ float dcn_post(float v, int co, int C, const f32x4& fl, float max_residue) {
    const int noff = (C / 3) * 2;
    if (co >= noff) return e2_fast_sigmoid(v);           // hardware exp2 / rcp: libm's tanhf / expf on 432 channels x 58 320
    const int which = (co * 2 >= noff) ? 2 : 0;          // pixels cost ~40 us of a 110 us launch at 720p
    return max_residue * e2_fast_tanh(v) + fl[which + ((co & 1) ? 0 : 1)];
}

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __builtin_bit_cast(float, (unsigned)b << 16); }

// x = hi + mid + lo with bf16 pieces (common.h, e2_split8).  Non-finite inputs give NaN.
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    e2_split8(v0, v1, hi, mid, lo);
}

// LDS-DMA piece through inline asm (the PP loop): hipcc neither counts the load nor waits for it -- with the builtin it put an
// s_waitcnt vmcnt(0) in front of the first ds_read behind every barrier (any LDS read may alias an outstanding LDS-DMA write), which
// exposes the whole DMA latency in the interval that issued it.  The loop's own vmcnt(0) in front of the barrier behind which the stage
// is first read is the only wait.  M0 = LDS byte address of the piece (wave-uniform), saved / restored (cdna_hip_programming.md 5.7).
typedef int pp_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ pp_i32x4 pp_rsrc_words(const void* base, unsigned bytes) {
    const unsigned long long b = (unsigned long long)base;
    pp_i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(b >> 32) & 0xFFFFu));
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}
__device__ __forceinline__ void pp_dma16(pp_i32x4 rsrc, unsigned lds_dst, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(rsrc), "s"(lds_dst), "s"(soff) : "memory");
}

// Two LDS stages: the DMA of step s+1 flies during the MFMAs of step s and is drained at the barrier that ends step s.
// (A third stage with counted vmcnt waits was measured slower on every layer -- it costs the second resident workgroup --
// and was removed, profiles/r02_bf16x_conv_microbench.txt.)
// S3 = true (3x3, stride 1, pad 1 only): the three horizontal taps of a kernel row share ONE A stage.  Output pixels are
// linear in (image, y, x), so the input pixel of tap kx is the input pixel of the centre tap of the linear neighbour
// m + kx - 1 whenever that neighbour is in the same image row: the stage holds BM + 2 rows (virtual pixels m0 - 1 ..
// m0 + BM, vertical offset ky - 1 applied), tap kx reads rows r + kx, and the two lanes whose neighbour would wrap into
// another image row (x = 0 with kx = 0, x = W - 1 with kx = 2) zero their operand instead.  A DMA per 9 taps: 3 stages of
// BM + 2 rows instead of 9 of BM; the weights are fetched per tap as before.  K walk: ky -> source -> block -> kx.
// MODE 1 (F32): the same kernel on fp32 operands (fp32 NHWC sources, fp32 packed weights, v_mfma_f32_32x32x2_f32 -- exact fp32):
// a K-step is then 32 channels (the same 128-byte rows, 16-byte chunks of 4 channels), everything else -- DMA, swizzle,
// stages, epilogue -- is shared.  Used by the fp32 path for its GEMM-shaped layers (token Linears, SoftSplit / SoftComp).
// MODE 2 (X3): fp32 operands on the bf16 matrix pipe by EXACT three-way splitting.  An fp32 number is the sum of three bf16
// numbers (8 + 8 + 8 significand bits: hi = x with its low 16 bits cleared, mid = (x - hi) likewise, lo = x - hi - mid, each
// difference exact in fp32), so a product a*b is the sum of nine bf16 products, each of them exact in the MFMA's fp32
// accumulator; the three smallest (mid*lo, lo*mid, lo*lo: <= 2^-22 of |a*b| together) are dropped, the other six are
// issued -- 6 bf16 MFMAs of 32 cycles for the work of 8 fp32 MFMAs of 64: 2.7x the matrix rate at fp32-level rounding
// (the measured error against an fp64 reference is that of the exact-fp32 kernel, tests/test_gpu_x3.py).  The A stage is
// MODE 1's (fp32 rows, same DMA, same swizzle): a lane reads its 8 consecutive channels as two 16-byte chunks and splits
// them in registers (44 VALU per fragment, shared by the 6 x TN MFMAs it feeds); the weights are split once, at packing
// time, into three bf16 planes: a B stage is [3 planes][4 k-octets][BN][8 bf16].
// PP (round 6, MODE 2 only, 8-wave tiles; tile codes 107 / 108): the two waves of a SIMD (w and w + 4) run ONE INTERVAL APART, locked by
// a barrier per interval (four per K-step): while one half of the workgroup multiplies a 16-channel unit (48 MFMAs + the B-plane reads
// on the 256x256 tile), the other half reads and splits the A fragments of its next unit (VALU + LDS), then they swap -- the regime
// MI355X_MICROARCH.md describes under "Two waves per SIMD": a SIMD's matrix pipe and its VALU run side by side only when its two waves
// are in complementary segments; with one barrier per step both waves split, then both multiply.  Same products in the same order:
// BIT-IDENTICAL to tiles 7 / 8 (tests/test_gpu_x3.py).  Measured (tools/probe/pp_ab.py, alternating rounds, one process): fc1 94.3 ->
// 85.0 us, qkv 70.5 -> 66.7, sc 253.0 -> 240.2; the interval timeline: tools/probe/pp_probe.py, profiles/r06_x3_gemm_pingpong.txt.
template <int BM, int BN, int WGM, int WGN, bool S3, int MODE, bool PP = false>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_bf16x_kernel(const ConvXParams p) {
#if defined(__HIP_DEVICE_COMPILE__)      // the LDS-DMA builtin takes an address_space(3) pointer the host pass cannot form
    constexpr bool F32 = MODE == 1, X3 = MODE == 2;
    static_assert(!PP || (X3 && !S3 && WGM * WGN == 8), "ping-pong halves: split-operand mode, 8 waves, no row-shift stages");
    constexpr int NT = 64 * WGM * WGN;
    constexpr int ESZ = MODE ? 4 : 2;                           // activation element size
    constexpr int CH = 16 / ESZ;                                // channels per 16-byte chunk
    constexpr int KC = 8 * CH;                                  // channels per K-step (128-byte rows)
    constexpr int TM = BM / (32 * WGM), TN = BN / (32 * WGN);
    constexpr int AR = S3 ? BM + 2 : BM;                        // rows of an A stage
    constexpr int B_CH = X3 ? 12 : 8;                           // 16-byte chunks per output column in a B stage
    constexpr int A_BYTES = AR * 128, B_BYTES = BN * B_CH * 16; // one K-step stage of each operand
    constexpr int A_IT = (AR * 8 + NT - 1) / NT, B_IT = (BN * B_CH + NT - 1) / NT;   // 16-byte DMA items per thread
    constexpr bool A_PART = (AR * 8) % NT != 0;                 // the last A iteration is partial (S3: 16 extra items)
    constexpr bool B_PART = (BN * B_CH) % NT != 0;              // (X3, 32-column tile: whole waves drop out)
    constexpr int R = TM * 32, CN = TN * 32;                    // a wave's output block
    // the epilogue parks the accumulators in LDS, the whole block at once or (256x256 tile) in two column halves
    // (three passes of one column tile for the 64 x 96 wave blocks of the 256x192 tile)
    constexpr int ES = (WGM * WGN * R * (CN + 4) * 4 <= 150 * 1024) ? 1 : (TN % 2 == 0 ? 2 : TN);
    constexpr int TNH = TN / ES, CNH = CN / ES, LDE = CNH + 4;  // a wave's epilogue region: R rows of LDE floats
    constexpr int EPI = WGM * WGN * R * LDE * 4;
    static_assert(TN % ES == 0 && EPI <= 160 * 1024, "epilogue region");
    constexpr int SMEM = 2 * (A_BYTES + B_BYTES) > EPI ? 2 * (A_BYTES + B_BYTES) : EPI;   // A stages first, then B stages
    constexpr unsigned OOB = 0xFFFFFFFFu;
    static_assert(TM >= 1 && TN >= 1 && (BM * 8) % NT == 0 && (BN * B_CH) % 64 == 0 && SMEM <= 160 * 1024, "tile");

    __shared__ __attribute__((aligned(1024))) unsigned char smem[SMEM];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);    // scalar: the LDS-DMA base (M0) is SALU work
    const int wm = wave / WGN, wn = wave % WGN;
    const int g = blockIdx.y;
    const int logical = xcd_remap(blockIdx.x, p.tilesM * p.tilesN);
    const int tile_m = logical / p.tilesN, tile_n = logical - tile_m * p.tilesN;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int HoWo = p.Ho * p.Wo;

    // ---- DMA bookkeeping.  The K loop must stay almost free of VALU / SALU work: every LDS-DMA piece that needs a dozen
    // address instructions costs as much issue time as the MFMAs it feeds (SQ_INSTS_VALU was 4.5 per MFMA with per-step
    // address arithmetic, profiles/r02_bf16x_pmc_raw.txt).  So a thread keeps ONE ready-made byte offset per A item, valid
    // for the current (tap, source) -- or the out-of-range sentinel for halo / tail rows, which the buffer bounds check
    // turns into zeros -- and the channel block inside the source advances through the instruction's SCALAR offset; the
    // weight offsets are constant and the K-step advances through the scalar offset as well.
    int a_pix[A_IT];                       // input pixel (linear) of tap (0, 0)
    unsigned a_msk[A_IT];                  // bit ky: row by + ky inside the image; bit 8 + kx: column bx + kx inside
    unsigned a_ch16[A_IT];                 // byte offset of the item's (swizzled) chunk inside a K-step
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int item = tid + it * NT;
        const int row = item >> 3, slot = item & 7;
        a_ch16[it] = (unsigned)((slot ^ ((row >> 1) & 7)) * 16);
        const int m = S3 ? m0 + row - 1 : m0 + row;             // S3: the stage row's virtual pixel
        const bool ok = m >= 0 && m < p.M;
        const int mm = ok ? m : 0;
        const int img = mm / HoWo;
        const int rem = mm - img * HoWo;
        const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
        const int by = oy * p.stride - p.pad;
        const int bx = S3 ? ox : ox * p.stride - p.padx;        // S3 stages the centre column; the tap shifts the READ
        unsigned msk = 0;
        for (int k = 0; k < p.KH; ++k) msk |= ((unsigned)(by + k) < (unsigned)p.H ? 1u : 0u) << k;
        for (int k = 0; k < (S3 ? 1 : p.KW); ++k) msk |= ((unsigned)(bx + k) < (unsigned)p.W ? 1u : 0u) << (8 + k);
        a_msk[it] = ok ? msk : 0u;
        a_pix[it] = (img * p.H + by) * p.W + bx;
    }
    unsigned b_off[B_IT];
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        const int item = tid + it * NT;
        const int koct = item / BN, n = item - koct * BN;
        b_off[it] = ((n0 + n) < p.Npad && koct < B_CH) ? (unsigned)((koct * p.Npad + n0 + n) * 16) : OOB;
    }
    const __amdgpu_buffer_rsrc_t wrsrc = make_rsrc(reinterpret_cast<const char*>(p.w) + (long long)g * p.wgroup_bytes, p.wgroup_bytes);
    const unsigned b_step = (unsigned)B_CH * (unsigned)p.Npad * 16u;   // packed-weight bytes per K-step

    // walk of the K-steps: tap (ky, kx) -> source s -> 64-channel block c0.  The per-source parameters are read from the
    // kernel arguments ONCE (indexing the argument arrays per step costs dependent scalar loads and a wait in the loop).
    const void* const sp0 = p.src[0]; const void* const sp1 = p.src[1]; const void* const sp2 = p.src[2]; const void* const sp3 = p.src[3];
    const unsigned sb0 = p.src_bytes[0], sb1 = p.src_bytes[1], sb2 = p.src_bytes[2], sb3 = p.src_bytes[3];
    const unsigned sl0 = (unsigned)p.ld[0] * ESZ, sl1 = (unsigned)p.ld[1] * ESZ, sl2 = (unsigned)p.ld[2] * ESZ, sl3 = (unsigned)p.ld[3] * ESZ;
    const unsigned sc0 = (unsigned)(p.coff[0] + g * p.cpg[0]) * ESZ, sc1 = (unsigned)(p.coff[1] + g * p.cpg[1]) * ESZ,
                   sc2 = (unsigned)(p.coff[2] + g * p.cpg[2]) * ESZ, sc3 = (unsigned)(p.coff[3] + g * p.cpg[3]) * ESZ;
    const int sg0 = p.cpg[0], sg1 = p.cpg[1], sg2 = p.cpg[2], sg3 = p.cpg[3];
    int ky = 0, kx = 0, s = 0, c0 = 0;
    int blk = 0;                                                  // S3: block index inside the kernel row's tap
    const void* cur_src = sp0;
    unsigned cur_bytes = sb0;
    unsigned cur_ld2 = sl0, cur_chan = sc0;
    int cur_cpg = sg0;
    unsigned a_off[A_IT];                                         // this (tap, source)'s byte offsets (block 0) or OOB
    auto retarget = [&]() {                                       // after ky / kx / s changed
        if (p.nsrc > 1) {
            cur_src = s == 0 ? sp0 : s == 1 ? sp1 : s == 2 ? sp2 : sp3;
            cur_bytes = s == 0 ? sb0 : s == 1 ? sb1 : s == 2 ? sb2 : sb3;
            cur_ld2 = s == 0 ? sl0 : s == 1 ? sl1 : s == 2 ? sl2 : sl3;
            cur_chan = s == 0 ? sc0 : s == 1 ? sc1 : s == 2 ? sc2 : sc3;
            cur_cpg = s == 0 ? sg0 : s == 1 ? sg1 : s == 2 ? sg2 : sg3;
        }
        const int kxe = S3 ? 0 : kx;
        const int tap = ky * p.W + kxe;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            const bool ok = ((a_msk[it] >> ky) & (a_msk[it] >> (8 + kxe)) & 1u) != 0;
            a_off[it] = ok ? (unsigned)(a_pix[it] + tap) * cur_ld2 + cur_chan + a_ch16[it] : OOB;
        }
    };
    retarget();
    unsigned char* const smem_b = smem + 2 * A_BYTES;
    // The whole next stage is issued BEFORE the MFMAs of the current one.  Spreading the pieces between the four MFMA groups
    // of the step (a quarter after each group's operand reads) was measured 5-10 % slower on every layer: the loads are
    // latency-exposed, the earliest possible issue wins (profiles/r02_bf16x_conv_microbench.txt).
    auto issue_a = [&](int stage) {
        if (p.dbg_noload & 16) return;
        unsigned char* sa = smem + stage * A_BYTES + wave * 1024;
        const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(cur_src, cur_bytes);
        const unsigned soff = (unsigned)c0 * (unsigned)ESZ;       // the channel block: scalar offset of the instruction
        if (c0 + KC <= cur_cpg) {
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
                if (!A_PART || it + 1 < A_IT || tid + it * NT < AR * 8)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_void*)(sa + it * NT * 16), 16, a_off[it], soff, 0, 0);
        } else {                                                  // the source's last, partial block: chunks past its channels are zeros
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
                if (!A_PART || it + 1 < A_IT || tid + it * NT < AR * 8)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_void*)(sa + it * NT * 16), 16,
                                                             (c0 + (int)(a_ch16[it] / ESZ) < cur_cpg) ? a_off[it] : OOB, soff, 0, 0);
        }
    };
    auto issue_b = [&](int stage, int step) {
        if (p.dbg_noload & 32) return;
        unsigned char* sb = smem_b + stage * B_BYTES + wave * 1024;
        const unsigned soff = (unsigned)step * b_step;
#pragma unroll
        for (int it = 0; it < B_IT; ++it)
            if (!B_PART || it + 1 < B_IT || tid + it * NT < BN * B_CH)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (lds_void*)(sb + it * NT * 16), 16, b_off[it], soff, 0, 0);
    };
    auto advance = [&]() -> bool {                                // next (source, block); past the last one: next tap / kernel row
        c0 += KC;
        ++blk;
        if (c0 < cur_cpg) return false;
        c0 = 0;
        ++s;
        if (s == p.nsrc) {
            s = 0;
            blk = 0;
            if (S3) ++ky;
            else { ++kx; if (kx == p.KW) { kx = 0; ++ky; } }
        }
        return true;                                              // the caller re-targets the A offsets
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TN; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tm][tn][r] = 0.f;

    const int i = lane & 31, h = lane >> 5;
    // LDS byte offsets of this lane's operand reads inside a stage (kk = 0); kk advances the chunk by 2
    int a_row[TM], b_rd[TN];
    bool okl[TM], okr[TM];                 // S3: this lane's output pixel has a left / right neighbour in its image row
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        a_row[tm] = (wm * TM + tm) * 32 + i;
        const int ox = (m0 + a_row[tm]) % p.Wo;
        okl[tm] = ox != 0;
        okr[tm] = ox != p.Wo - 1;
    }
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) b_rd[tn] = ((wn * TN + tn) * 32 + i) * 16;

    auto compute = [&](const unsigned char* st, const unsigned char* stb, int kxs) {
        int a_rd[TM], a_key[TM];
        bool zero[TM];
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
            const int row = a_row[tm] + (S3 ? kxs : 0);
            a_rd[tm] = row * 128;
            a_key[tm] = (row >> 1) & 7;
            zero[tm] = S3 && ((kxs == 0 && !okl[tm]) || (kxs == 2 && !okr[tm]));
        }
        if constexpr (X3) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int q = 2 * kk + h;                        // this half-wave's k-octet of the step (8 channels = chunks 2q, 2q + 1)
                bf16x8 ah[TM], am[TM], al[TM];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    split8(*reinterpret_cast<const f32x4*>(st + a_rd[tm] + (((2 * q) ^ a_key[tm]) << 4)),
                           *reinterpret_cast<const f32x4*>(st + a_rd[tm] + (((2 * q + 1) ^ a_key[tm]) << 4)), ah[tm], am[tm], al[tm]);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(stb + b_rd[tn] + q * (BN * 16));
                    const bf16x8 bm = *reinterpret_cast<const bf16x8*>(stb + b_rd[tn] + (4 + q) * (BN * 16));
                    const bf16x8 bl = *reinterpret_cast<const bf16x8*>(stb + b_rd[tn] + (8 + q) * (BN * 16));
                    // smallest terms first; the TM accumulators of a term are independent
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[tm], bh, acc[tm][tn], 0, 0, 0);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bl, acc[tm][tn], 0, 0, 0);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[tm], bm, acc[tm][tn], 0, 0, 0);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am[tm], bh, acc[tm][tn], 0, 0, 0);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bm, acc[tm][tn], 0, 0, 0);
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[tm], bh, acc[tm][tn], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if constexpr (F32) {
                // chunk 2 kk + h holds 4 consecutive k of the row / column: MFMA #e multiplies k = 4 (2 kk + h) + e of both halves
                f32x4 a[TM], b[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
                    a[tm] = *reinterpret_cast<const f32x4*>(st + a_rd[tm] + (((2 * kk + h) ^ a_key[tm]) << 4));
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    b[tn] = *reinterpret_cast<const f32x4*>(stb + b_rd[tn] + (2 * kk + h) * (BN * 16));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[tm][e], b[tn][e], acc[tm][tn], 0, 0, 0);
            } else {
                bf16x8 a[TM], b[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) {
                    a[tm] = *reinterpret_cast<const bf16x8*>(st + a_rd[tm] + (((2 * kk + h) ^ a_key[tm]) << 4));
                    if (S3) {
                        u32x4 q = __builtin_bit_cast(u32x4, a[tm]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) q[e] = zero[tm] ? 0u : q[e];
                        a[tm] = __builtin_bit_cast(bf16x8, q);
                    }
                }
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    b[tn] = *reinterpret_cast<const bf16x8*>(stb + b_rd[tn] + (2 * kk + h) * (BN * 16));
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[tm], b[tn], acc[tm][tn], 0, 0, 0);
            }
        }
        }
    };
    const bool tap_packed = !S3 && p.tp_cq != 0;          // its first A stage is issued by the packed branch below
    if (!tap_packed) issue_a(0);
    issue_b(0, 0);
    if (!tap_packed) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if constexpr (PP) {
        // (host: PP tiles reject tap-packed weights)
        // Unit of the exchange = HALF a K-step (one k-octet pair kk: 16 channels): the fragments of one unit are 3 TM operand
        // registers-of-4 -- with those of a whole step the 256x256 tile spilled 350 registers.  Four intervals per step, a barrier
        // behind each; side by side in time:   leading half:  P(kk0)       M(kk0)  P(kk1)  M(kk1)
        //                                     trailing half: M(kk1, s-1)  P(kk0)  M(kk0)  P(kk1)
        const bool trail = wave >= 4;                              // wave-uniform: waves w and w + 4 share a SIMD
        bf16x8 fa[TM][3];                                          // the unit's A fragments: [row tile][hi, mid, lo]
        auto pp_bar = [&]() __attribute__((always_inline)) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        };
        auto prep = [&](const unsigned char* st, const int kk) __attribute__((always_inline)) {
            const int q = 2 * kk + h;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int row = a_row[tm];
                const int rd = row * 128, key = (row >> 1) & 7;
                split8(*reinterpret_cast<const f32x4*>(st + rd + (((2 * q) ^ key) << 4)),
                       *reinterpret_cast<const f32x4*>(st + rd + (((2 * q + 1) ^ key) << 4)), fa[tm][0], fa[tm][1], fa[tm][2]);
            }
        };
        // the B planes of column tile tn are read one tile ahead of their six TM MFMAs (two fragment sets; the fences keep hipcc from
        // hoisting every read of the phase to its top)
        auto mma = [&](const unsigned char* stb, const int kk) __attribute__((always_inline)) {
            const int q = 2 * kk + h;
            bf16x8 bq[2][3];
            auto rd = [&](int tn, bf16x8 (&b)[3]) __attribute__((always_inline)) {
                b[0] = *reinterpret_cast<const bf16x8*>(stb + b_rd[tn] + q * (BN * 16));
                b[1] = *reinterpret_cast<const bf16x8*>(stb + b_rd[tn] + (4 + q) * (BN * 16));
                b[2] = *reinterpret_cast<const bf16x8*>(stb + b_rd[tn] + (8 + q) * (BN * 16));
            };
            rd(0, bq[0]);
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) {
                if (tn + 1 < TN) rd(tn + 1, bq[(tn + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);      // the reads stay HERE, twelve MFMAs ahead of their use (hipcc sank them to two MFMAs ahead)
                const bf16x8 bh = bq[tn & 1][0], bm = bq[tn & 1][1], bl = bq[tn & 1][2];
                // the order of compute(): smallest terms first, the TM accumulators of a term are independent
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][2], bh, acc[tm][tn], 0, 0, 0);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], bl, acc[tm][tn], 0, 0, 0);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][1], bm, acc[tm][tn], 0, 0, 0);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][1], bh, acc[tm][tn], 0, 0, 0);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], bm, acc[tm][tn], 0, 0, 0);
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[tm][0], bh, acc[tm][tn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // ONE instruction stream for both halves -- P(kk0) | M(kk0) | P(kk1) | M(kk1) per step, a barrier behind each -- and the trailing
        // half runs it ONE INTERVAL LATER (an extra barrier in front of its loop, one behind the leading half's).  (With the roles
        // switched by branches around the MFMAs -- `if (lead) mma else prep` -- hipcc no longer kept the 128 accumulator registers in
        // place across the joins: 414 spilled registers on the 256x256 tile; with advance / retarget / issue wrapped in one more lambda
        // it copied the kernel arguments to scratch.)  In global intervals t (leading half: P(2s) at t = 4s, trailing half one later):
        // stage s - 1's last reader is the trailing M(2s - 1) at t = 4s, stage s + 1's first reader the leading P(2s + 2) at t = 4s + 4.
        // The trailing half issues its share of stage s + 1 at the top of its step (t = 4s + 1, its P(2s) interval), the leading half
        // in its P(2s + 1) interval (t = 4s + 2) -- never inside a multiply interval, where ten asm pieces cost ~800 cycles -- and both
        // wait for their pieces in front of the barrier that ends t = 4s + 3 (leading: its fourth of the step, trailing: its third).
        const unsigned smem_lds = (unsigned)(unsigned long long)(lds_void*)smem;
        const pp_i32x4 wrsrc_w = pp_rsrc_words(reinterpret_cast<const char*>(p.w) + (long long)g * p.wgroup_bytes, p.wgroup_bytes);
        auto pp_issue = [&](int stage, int step) __attribute__((always_inline)) {       // issue_a(stage) + issue_b(stage, step), through asm
            const pp_i32x4 arsrc_w = pp_rsrc_words(cur_src, cur_bytes);
            const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(c0 * ESZ);
            const unsigned la = smem_lds + (unsigned)(stage * A_BYTES + wave * 1024);
            const bool whole = c0 + KC <= cur_cpg;
#pragma unroll
            for (int it = 0; it < A_IT; ++it)
                pp_dma16(arsrc_w, (unsigned)__builtin_amdgcn_readfirstlane((int)(la + it * NT * 16)),
                         (whole || c0 + (int)(a_ch16[it] / ESZ) < cur_cpg) ? a_off[it] : OOB, soff);
            const unsigned lb = smem_lds + (unsigned)(2 * A_BYTES + stage * B_BYTES + wave * 1024);
            const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)step * b_step));
#pragma unroll
            for (int it = 0; it < B_IT; ++it)
                if (!B_PART || it + 1 < B_IT || tid + it * NT < BN * B_CH)
                    pp_dma16(wrsrc_w, (unsigned)__builtin_amdgcn_readfirstlane((int)(lb + it * NT * 16)), b_off[it], sb);
        };
        if (trail) pp_bar();
        for (int step = 0; step < p.nsteps; ++step) {
            const int cur = step & 1;
            const unsigned char* const sA = smem + cur * A_BYTES;
            const unsigned char* const sB = smem_b + cur * B_BYTES;
            if (trail) { if (step + 1 < p.nsteps) { if (advance()) retarget(); pp_issue((step & 1) ^ 1, step + 1); } }
            prep(sA, 0);
            pp_bar();
            mma(sB, 0);
            pp_bar();
            prep(sA, 1);
            if (!trail) { if (step + 1 < p.nsteps) { if (advance()) retarget(); pp_issue((step & 1) ^ 1, step + 1); } }
            if (trail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pp_bar();
            mma(sB, 1);
            if (!trail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            pp_bar();
        }
        if (!trail) pp_bar();
    } else
    if constexpr (!S3) {
        if (p.tp_cq) {
            // Tap-packed K-steps: a source of fewer than 64 channels (bf16; fp32: 32 channels per step, 4 per chunk -- the same
            // bytes) fills only cq = cpg / 8 of the 8 chunks of a K-step, so the
            // (tap, channel chunk) pairs are laid out as ONE stream of chunks, tap-major, cut into K-steps of 8 -- a step then
            // carries 8 / cq taps (8 channels: 8 taps, 16: 4, 32: 2, 40: 1.6) instead of one zero-padded tap: SPyNet's 7x7 layers
            // on 8 / 16 / 32 channels (49 -> 7 / 13 / 25 steps), the encoder's first layer, the FFN's second Linear read as the 7x7
            // stride-3 convolution it is over the 40-channel folded tensor (49 -> 31 steps).  Chunk L = 8 step + c of the stream =
            // tap L / cq, channel chunk L % cq; the lanes of a row then fetch from different taps, so the byte offsets are formed
            // per step (a dozen VALU per item -- cheap against the steps it removes).
            const int cq = p.tp_cq, KK = p.KH * p.KW;
            int t_c[A_IT];
#pragma unroll
            for (int it = 0; it < A_IT; ++it) t_c[it] = (int)(a_ch16[it] >> 4);
            const __amdgpu_buffer_rsrc_t arsrc = make_rsrc(sp0, sb0);
            auto issue_a_packed = [&](int stage, int step) {
                unsigned char* sa = smem + stage * A_BYTES + wave * 1024;
#pragma unroll
                for (int it = 0; it < A_IT; ++it) {
                    const int L = 8 * step + t_c[it];
                    const int tap = cq == 1 ? L : (int)__umulhi((unsigned)L, p.tp_magic_cq);
                    const int cc = L - tap * cq;
                    const int tky = (int)__umulhi((unsigned)tap, p.tp_magic), tkx = tap - tky * p.KW;
                    const bool ok = tap < KK && ((a_msk[it] >> tky) & (a_msk[it] >> (8 + tkx)) & 1u) != 0;
                    const unsigned off = ok ? (unsigned)(a_pix[it] + tky * p.W + tkx) * sl0 + sc0 + (unsigned)cc * 16u : OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(arsrc, (lds_void*)(sa + it * NT * 16), 16, off, 0, 0, 0);
                }
            };
            issue_a_packed(0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int step = 0; step < p.nsteps; ++step) {
                const int cur = step & 1;
                if (step + 1 < p.nsteps) {
                    issue_a_packed(cur ^ 1, step + 1);
                    issue_b(cur ^ 1, step + 1);
                }
                compute(smem + cur * A_BYTES, smem_b + cur * B_BYTES, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
        } else
        for (int step = 0; step < p.nsteps; ++step) {
            const int cur = step & 1;
            const bool more = step + 1 < p.nsteps && !(p.dbg_noload & 1);
            if (more) {
                if (advance()) retarget();
                issue_a(cur ^ 1);
                issue_b(cur ^ 1, step + 1);
            }
            compute(smem + cur * A_BYTES, smem_b + cur * B_BYTES, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next stage has landed (this wave's share)
            __syncthreads();                                        // ... everybody's, and this stage's readers are done
        }
    } else if constexpr (!S3) {
        for (int step = 0; step < p.nsteps; ++step) {
            const int cur = step & 1;
            const bool more = step + 1 < p.nsteps && !(p.dbg_noload & 1);
            if (more) {
                if (advance()) retarget();
                issue_a(cur ^ 1);
                issue_b(cur ^ 1, step + 1);
            }
            compute(smem + cur * A_BYTES, smem_b + cur * B_BYTES, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the next stage has landed (this wave's share)
            __syncthreads();                                        // ... everybody's, and this stage's readers are done
        }
    } else {
        const int spt = p.nsteps / 9;                               // blocks per tap
        int acur = 0, kxs = 0;
        for (int step = 0; step < p.nsteps; ++step) {
            const bool more = step + 1 < p.nsteps && !(p.dbg_noload & 1);
            int kxn = kxs + 1;
            bool newblk = false;                                    // the next sub-step opens a new (kernel row, block): new A stage
            if (kxn == 3) {
                kxn = 0;
                if (more) { if (advance()) retarget(); newblk = true; }
            }
            if (more) {
                if (newblk) issue_a(acur ^ 1);
                issue_b((step + 1) & 1, (ky * 3 + kxn) * spt + blk);
            }
            compute(smem + acur * A_BYTES, smem_b + (step & 1) * B_BYTES, kxs);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (kxn == 0) acur ^= 1;
            kxs = kxn;
        }
    }

    if (p.dbg_noload & 2) {                // measurement aid: no epilogue (keeps the accumulators alive through one store)
        float sacc = 0.f;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[tm][tn][r];
        if (sacc == 123.456f) reinterpret_cast<float*>(p.dst)[0] = sacc;
        return;
    }
    // ---- epilogue through LDS
    // where output row m lives: row m of dst (default), or -- with an output scatter (osy > 0: the phase convolutions of the
    // gather-form SoftComp write every third pixel of the folded image) -- pixel (oy*osy + opy, ox*osx + opx) of its image;
    // mr: the residual's row (the same, or the pixel alone when one residual image is broadcast over the batch)
    auto out_row = [&](long long m, long long& mr) -> long long {
        long long mo = m;
        mr = m;
        if (p.osy | p.res_bcast) {
            const int img = (int)(m / HoWo);
            const int rem = (int)(m - (long long)img * HoWo);
            if (p.osy) {
                const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
                const int pix = (oy * p.osy + p.opy) * p.oW + ox * p.osx + p.opx;
                mo = (long long)img * p.oH * p.oW + pix;
                mr = p.res_bcast ? pix : mo;
            } else {
                mr = rem;
            }
        }
        return mo;
    };
    float* E = reinterpret_cast<float*>(smem) + wave * (R * LDE);
    constexpr int LPR = CNH / 8;           // lanes per row (8 channels each)
    constexpr int RPP = 64 / LPR;          // rows per pass
    const int col0 = (lane % LPR) * 8;
#pragma unroll
    for (int eh = 0; eh < ES; ++eh) {
    if (eh) __syncthreads();               // the previous half has been read out
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int tn = 0; tn < TNH; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                E[(tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * LDE + tn * 32 + i] = acc[tm][eh * TNH + tn][r];
    __syncthreads();
    const int n = n0 + wn * CN + eh * CNH + col0;     // first of this lane's 8 channels inside the group
    // Fast path, decided per wave (every term is wave-uniform): the wave's whole R x CN block lies inside the problem, the
    // activation is none / ReLU / LeakyReLU and every tensor is 16-byte addressable per 8 channels.  Straight-line code, no
    // per-lane predicates: the general path below masks every element (channel tails, row tails, scalar stores) and costs
    // ~3x the instructions -- on a 512-deep token GEMM that was as much as the K loop (profiles/r02_bf16x_ablation.txt).
    const bool fast = n0 + wn * CN + (eh + 1) * CNH <= p.Cout_g && m0 + wm * R + R <= p.M && p.act <= E2FGVI_ACT_LRELU && !p.dst_nchw &&
                      ((p.dst_ld | p.dst_coff | (g * p.Cout_g)) & 7) == 0 && (!p.res || ((p.res_ld | p.res_coff) & 7) == 0) &&
                      (!p.dst2 || ((p.dst2_ld | p.dst2_coff) & 7) == 0) && !(p.dbg_noload & 8);
    if (fast) {
        const int co = g * p.Cout_g + n;
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = b0;
        if (p.bias) { b0 = *reinterpret_cast<const f32x4*>(p.bias + co); b1 = *reinterpret_cast<const f32x4*>(p.bias + co + 4); }
        const float neg = p.act == E2FGVI_ACT_RELU ? 0.f : (p.act == E2FGVI_ACT_LRELU ? p.slope : 1.f);
#pragma unroll 1
        for (int ps = 0; ps < R / RPP; ++ps) {
            const int row = ps * RPP + lane / LPR;
            long long mr;
            const long long m = out_row(m0 + wm * R + row, mr);
            f32x4 v0 = *reinterpret_cast<const f32x4*>(E + row * LDE + col0) + b0;
            f32x4 v1 = *reinterpret_cast<const f32x4*>(E + row * LDE + col0 + 4) + b1;
            if (p.res) {
                const long long ro = mr * p.res_ld + p.res_coff + co;
                if (p.res_bf16) {
                    const u32x4 q = *reinterpret_cast<const u32x4*>(reinterpret_cast<const unsigned short*>(p.res) + ro);
                    v0[0] += __builtin_bit_cast(float, q[0] << 16); v0[1] += __builtin_bit_cast(float, q[0] & 0xFFFF0000u);
                    v0[2] += __builtin_bit_cast(float, q[1] << 16); v0[3] += __builtin_bit_cast(float, q[1] & 0xFFFF0000u);
                    v1[0] += __builtin_bit_cast(float, q[2] << 16); v1[1] += __builtin_bit_cast(float, q[2] & 0xFFFF0000u);
                    v1[2] += __builtin_bit_cast(float, q[3] << 16); v1[3] += __builtin_bit_cast(float, q[3] & 0xFFFF0000u);
                } else {
                    const float* rp = reinterpret_cast<const float*>(p.res) + ro;
                    v0 = v0 + *reinterpret_cast<const f32x4*>(rp);
                    v1 = v1 + *reinterpret_cast<const f32x4*>(rp + 4);
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {              // none (neg = 1), ReLU (0), LeakyReLU (slope)
                v0[c] = v0[c] > 0.f ? v0[c] : v0[c] * neg;
                v1[c] = v1[c] > 0.f ? v1[c] : v1[c] * neg;
            }
            if (p.split_plane && co >= p.split_from) {
                // the K / V columns of a qkv Linear: the operand planes of the split-operand attention, not the fp32 rows
                bf16x8 sh, sm, sl;
                e2_split8(v0, v1, sh, sm, sl);
                __bf16* o = p.dst2 + m * p.dst2_ld + p.dst2_coff + (co - p.split_from);
                *reinterpret_cast<bf16x8*>(o) = sh;
                *reinterpret_cast<bf16x8*>(o + p.split_plane) = sm;
                *reinterpret_cast<bf16x8*>(o + 2 * p.split_plane) = sl;
                continue;
            }
            bf16x8 hv = {(__bf16)v0[0], (__bf16)v0[1], (__bf16)v0[2], (__bf16)v0[3], (__bf16)v1[0], (__bf16)v1[1], (__bf16)v1[2], (__bf16)v1[3]};
            const long long dof = m * p.dst_ld + p.dst_coff + co;
            if (p.dst_bf16) {
                *reinterpret_cast<bf16x8*>(reinterpret_cast<__bf16*>(p.dst) + dof) = hv;
            } else {
                float* o = reinterpret_cast<float*>(p.dst) + dof;
                *reinterpret_cast<f32x4*>(o) = v0;
                *reinterpret_cast<f32x4*>(o + 4) = v1;
            }
            if (p.dst2 && !p.split_plane) *reinterpret_cast<bf16x8*>(p.dst2 + m * p.dst2_ld + p.dst2_coff + co) = hv;
        }
        continue;
    }
    if (n < p.Cout_g) {
        const int co = g * p.Cout_g + n;
        const bool full = n + 7 < p.Cout_g;
        float bv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) bv[c] = (p.bias && (full || n + c < p.Cout_g)) ? p.bias[co + c] : 0.f;
        const bool vec_d = full && ((p.dst_ld | p.dst_coff | co) & 7) == 0;
        const bool vec_r = full && p.res && ((p.res_ld | p.res_coff | co) & 7) == 0;
        const bool vec_2 = full && p.dst2 && ((p.dst2_ld | p.dst2_coff | co) & 7) == 0;
        // NOT unrolled: the pass body is a few hundred instructions with run-time variants (dtypes, residual, activation);
        // unrolled eight times the epilogue was tens of KB of cold straight-line code per workgroup and cost as much as a
        // 512-deep K loop (instruction fetch), profiles/r02_bf16x_ablation.txt
#pragma unroll 1
        for (int ps = 0; ps < R / RPP; ++ps) {
            const int row = ps * RPP + lane / LPR;
            const int m_in = m0 + wm * R + row;
            if (m_in >= p.M) continue;
            long long mr;
            const long long m = out_row(m_in, mr);
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(E + row * LDE + col0);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(E + row * LDE + col0 + 4);
            float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] += bv[c];
            if (p.act == E2FGVI_ACT_DCNPOST) {
                const f32x4 fl = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.res) + (long long)m_in * 4);
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = dcn_post(v[c], co + c, p.Cout, fl, p.slope);
            } else {
                if (p.res) {
                    const long long ro = mr * p.res_ld + p.res_coff + co;
                    if (p.res_bf16) {
                        const unsigned short* rp = reinterpret_cast<const unsigned short*>(p.res) + ro;
                        if (vec_r) {
                            const u32x4 q = *reinterpret_cast<const u32x4*>(rp);
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                v[2 * c] += __builtin_bit_cast(float, q[c] << 16);
                                v[2 * c + 1] += __builtin_bit_cast(float, q[c] & 0xFFFF0000u);
                            }
                        } else {
#pragma unroll
                            for (int c = 0; c < 8; ++c) if (full || n + c < p.Cout_g) v[c] += bf16_bits_to_f32(rp[c]);
                        }
                    } else {
                        const float* rp = reinterpret_cast<const float*>(p.res) + ro;
                        if (vec_r) {
                            const f32x4 q0 = *reinterpret_cast<const f32x4*>(rp), q1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
                            for (int c = 0; c < 4; ++c) { v[c] += q0[c]; v[4 + c] += q1[c]; }
                        } else {
#pragma unroll
                            for (int c = 0; c < 8; ++c) if (full || n + c < p.Cout_g) v[c] += rp[c];
                        }
                    }
                }
#pragma unroll
                for (int c = 0; c < 8; ++c) v[c] = apply_act(v[c], p.act, p.slope);
            }
            if (p.split_plane && co >= p.split_from) {       // (the launcher guarantees Cout % 8 == 0 and 16-byte addressable planes)
                const f32x4 s0 = {v[0], v[1], v[2], v[3]}, s1 = {v[4], v[5], v[6], v[7]};
                bf16x8 sh, sm, sl;
                e2_split8(s0, s1, sh, sm, sl);
                __bf16* o = p.dst2 + (long long)m * p.dst2_ld + p.dst2_coff + (co - p.split_from);
                *reinterpret_cast<bf16x8*>(o) = sh;
                *reinterpret_cast<bf16x8*>(o + p.split_plane) = sm;
                *reinterpret_cast<bf16x8*>(o + 2 * p.split_plane) = sl;
                continue;
            }
            bf16x8 hv;
#pragma unroll
            for (int c = 0; c < 8; ++c) hv[c] = (__bf16)v[c];
            const long long dof = (long long)m * p.dst_ld + p.dst_coff + co;
            if ((p.dbg_noload & 8) && v[0] != 123.456f) continue;          // measurement aid: everything but the global stores
            if (p.dst_nchw) {                      // plain fp32 NCHW [N,Cout,Ho,Wo] (the decoder's last layer: 3 channels)
                const int img = m_in / HoWo, rem = m_in - img * HoWo;
                float* o = reinterpret_cast<float*>(p.dst) + ((long long)img * p.Cout + co) * HoWo + rem;
#pragma unroll
                for (int c = 0; c < 8; ++c) if (full || n + c < p.Cout_g) o[(long long)c * HoWo] = v[c];
            } else if (p.dst_bf16) {
                __bf16* o = reinterpret_cast<__bf16*>(p.dst) + dof;
                if (vec_d) *reinterpret_cast<bf16x8*>(o) = hv;
                else {
#pragma unroll
                    for (int c = 0; c < 8; ++c) if (full || n + c < p.Cout_g) o[c] = hv[c];
                }
            } else {
                float* o = reinterpret_cast<float*>(p.dst) + dof;
                if (vec_d) {
                    f32x4 w0 = {v[0], v[1], v[2], v[3]}, w1 = {v[4], v[5], v[6], v[7]};
                    *reinterpret_cast<f32x4*>(o) = w0;
                    *reinterpret_cast<f32x4*>(o + 4) = w1;
                } else {
#pragma unroll
                    for (int c = 0; c < 8; ++c) if (full || n + c < p.Cout_g) o[c] = v[c];
                }
            }
            if (p.dst2 && !p.split_plane) {
                __bf16* o = p.dst2 + (long long)m * p.dst2_ld + p.dst2_coff + co;
                if (vec_2) *reinterpret_cast<bf16x8*>(o) = hv;
                else {
#pragma unroll
                    for (int c = 0; c < 8; ++c) if (full || n + c < p.Cout_g) o[c] = hv[c];
                }
            }
        }
    }
    }   // eh
#endif
}

struct PackX {
    int Cout, groups, KH, KW, nsrc;
    int cpg[E2FGVI_MAX_SRC];
    int Cout_g, Npad, Cin_g, steps_per_tap;
    int kc, ch;                           // channels per K-step (64 bf16 / 32 fp32) and per 16-byte chunk (8 / 4)
    int tp_cq;                            // tap-packed steps: chunks per tap (0 = off)
    long long total, wgroup_elems;        // elements
};

bool geometry_x(int Cout, int groups, int KH, int KW, int nsrc, const int32_t* cpg, PackX* q, bool f32 = false) {
    q->kc = f32 ? 32 : 64;
    q->ch = f32 ? 4 : 8;
    q->tp_cq = 0;
    if (Cout <= 0 || groups <= 0 || Cout % groups || KH <= 0 || KW <= 0 || nsrc < 1 || nsrc > E2FGVI_MAX_SRC) return false;
    q->Cout = Cout; q->groups = groups; q->KH = KH; q->KW = KW; q->nsrc = nsrc;
    q->Cout_g = Cout / groups;
    q->Npad = round_up(q->Cout_g, 32);
    q->Cin_g = 0;
    q->steps_per_tap = 0;
    for (int s = 0; s < E2FGVI_MAX_SRC; ++s) q->cpg[s] = 0;
    for (int s = 0; s < nsrc; ++s) {
        if (cpg[s] <= 0 || cpg[s] % q->ch) return false;
        q->cpg[s] = cpg[s];
        q->Cin_g += cpg[s];
        q->steps_per_tap += cdiv(cpg[s], q->kc);
    }
    q->wgroup_elems = (long long)KH * KW * q->steps_per_tap * q->kc * q->Npad;
    q->total = q->wgroup_elems * groups;
    return true;
}

// the weight that multiplies element e of 16-byte activation chunk koct of K-step `step` (group g, output column n)
__device__ __forceinline__ float pack_value(const float* __restrict__ w, const PackX& p, int g, int step, int koct, int n, int e) {
    if (p.tp_cq) {                        // tap-packed: chunk L = 8 step + koct of the stream = tap L / cq, channels (L % cq) * 8 ...
        const int L = 8 * step + koct;
        const int tapk = L / p.tp_cq;
        const int chk = (L - tapk * p.tp_cq) * p.ch + e;
        if (tapk < p.KH * p.KW && chk < p.cpg[0] && n < p.Cout_g)
            return w[((long long)n * p.Cin_g + chk) * (p.KH * p.KW) + tapk];
        return 0.f;
    }
    const int tap = step / p.steps_per_tap;
    int blk = step - tap * p.steps_per_tap;
    int s = 0, prefix = 0;
    while (blk >= (p.cpg[s] + p.kc - 1) / p.kc) { blk -= (p.cpg[s] + p.kc - 1) / p.kc; prefix += p.cpg[s]; ++s; }
    const int c = blk * p.kc + koct * p.ch + e;
    if (c < p.cpg[s] && n < p.Cout_g)
        return w[((long long)(g * p.Cout_g + n) * p.Cin_g + prefix + c) * (p.KH * p.KW) + tap];
    return 0.f;
}

// packed layout [group][K-step][8 chunks][Npad][ch elements]
template <typename T>
__global__ void pack_conv_weight_x_kernel(const float* __restrict__ w, T* __restrict__ wp, const PackX p) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.total) return;
    const int g = (int)(idx / p.wgroup_elems);
    long long rem = idx - (long long)g * p.wgroup_elems;
    const int e = (int)(rem % p.ch);
    rem /= p.ch;
    const int n = (int)(rem % p.Npad);
    rem /= p.Npad;
    const int koct = (int)(rem & 7);
    const int step = (int)(rem >> 3);
    wp[idx] = (T)pack_value(w, p, g, step, koct, n, e);
}

// X3 packing: the fp32 geometry (32 channels per K-step, 4 per activation chunk), every weight split into three bf16 pieces
// (hi / mid by clearing the low 16 bits of the value / of the exact remainder, lo = what is left: the sum is the weight,
// bit for bit).  Layout [group][K-step][plane hi, mid, lo][4 k-octets][Npad][8 bf16]; one thread per fp32 weight.
__global__ void pack_conv_weight_x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, const PackX p) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.total) return;
    const int g = (int)(idx / p.wgroup_elems);
    long long rem = idx - (long long)g * p.wgroup_elems;
    const int e = (int)(rem & 3);
    rem >>= 2;
    const int n = (int)(rem % p.Npad);
    rem /= p.Npad;
    const int koct = (int)(rem & 7);
    const int step = (int)(rem >> 3);
    const float v = pack_value(w, p, g, step, koct, n, e);
    const unsigned xb = __builtin_bit_cast(unsigned, v);
    const float r = v - __builtin_bit_cast(float, xb & 0xFFFF0000u);
    const unsigned rb = __builtin_bit_cast(unsigned, r);
    const float r2 = r - __builtin_bit_cast(float, rb & 0xFFFF0000u);
    const long long plane = 4LL * p.Npad * 8;
    unsigned short* o = wp + (long long)g * p.wgroup_elems * 3 + (long long)step * 3 * plane +
                        ((long long)(koct >> 1) * p.Npad + n) * 8 + (koct & 1) * 4 + e;
    o[0] = (unsigned short)(xb >> 16);
    o[plane] = (unsigned short)(rb >> 16);
    o[2 * plane] = (unsigned short)(__builtin_bit_cast(unsigned, r2) >> 16);
}

template <int BM, int BN, int WGM, int WGN, bool S3, bool PP = false>
int launch_x(ConvXParams& p, int groups, hipStream_t st, int mode = 0) {
    p.tilesM = cdiv(p.M, BM);
    p.tilesN = cdiv(p.Cout_g, BN);
    if constexpr (PP) {
        if (mode != 2 || p.tp_cq) {
            e2fgvi_set_error("conv2d_x: the ping-pong tiles (107, 108) are for the split-operand mode without tap-packed weights");
            return E2FGVI_EUNSUP;
        }
        hipLaunchKernelGGL((conv_bf16x_kernel<BM, BN, WGM, WGN, false, 2, true>), dim3(p.tilesM * p.tilesN, groups, 1), dim3(64 * WGM * WGN), 0, st, p);
        E2_LAUNCH_CHECK("conv2d_x");
        return 0;
    } else {
    if (S3 && !(p.KH == 3 && p.KW == 3 && p.stride == 1 && p.pad == 1)) {
        e2fgvi_set_error("conv2d_bf16x: the row-shift tiles (11..17) are for 3x3 stride-1 pad-1 layers");
        return E2FGVI_EINVAL;
    }
    if (mode) {
        if constexpr (!S3) {
            if (mode == 2)
                hipLaunchKernelGGL((conv_bf16x_kernel<BM, BN, WGM, WGN, false, 2>), dim3(p.tilesM * p.tilesN, groups, 1), dim3(64 * WGM * WGN), 0, st, p);
            else
                hipLaunchKernelGGL((conv_bf16x_kernel<BM, BN, WGM, WGN, false, 1>), dim3(p.tilesM * p.tilesN, groups, 1), dim3(64 * WGM * WGN), 0, st, p);
        } else {
            e2fgvi_set_error("conv2d_f32x: the row-shift tiles (11..18) take bf16 operands only");
            return E2FGVI_EUNSUP;
        }
    } else {
        hipLaunchKernelGGL((conv_bf16x_kernel<BM, BN, WGM, WGN, S3, 0>), dim3(p.tilesM * p.tilesN, groups, 1), dim3(64 * WGM * WGN), 0, st, p);
    }
    E2_LAUNCH_CHECK("conv2d_x");
    return 0;
    }
}

}  // namespace

extern "C" int64_t e2fgvi_packed_conv_weight_bf16x_size(int32_t Cout, int32_t groups, int32_t KH, int32_t KW, int32_t nsrc,
                                                        const int32_t* src_cpg) {
    PackX q;
    if (!src_cpg || !geometry_x(Cout, groups, KH, KW, nsrc, src_cpg, &q)) {
        e2fgvi_set_error("packed_conv_weight_bf16x_size: bad geometry (channels per source must be multiples of 8)");
        return E2FGVI_EINVAL;
    }
    return q.total;
}

extern "C" int e2fgvi_pack_conv_weight_bf16x(const float* w, void* wpacked, int32_t Cout, int32_t groups, int32_t KH,
                                             int32_t KW, int32_t nsrc, const int32_t* src_cpg, void* stream) {
    PackX q;
    E2_REQUIRE(w && wpacked && src_cpg, E2FGVI_EINVAL, "pack_conv_weight_bf16x: null pointer");
    E2_REQUIRE(geometry_x(Cout, groups, KH, KW, nsrc, src_cpg, &q), E2FGVI_EINVAL, "pack_conv_weight_bf16x: bad geometry");
    hipLaunchKernelGGL(pack_conv_weight_x_kernel<__bf16>, dim3((unsigned)cdiv64(q.total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, (__bf16*)wpacked, q);
    E2_LAUNCH_CHECK("pack_conv_weight_bf16x");
    return 0;
}

extern "C" int64_t e2fgvi_packed_conv_weight_f32x_size(int32_t Cout, int32_t groups, int32_t KH, int32_t KW, int32_t nsrc,
                                                       const int32_t* src_cpg) {
    PackX q;
    if (!src_cpg || !geometry_x(Cout, groups, KH, KW, nsrc, src_cpg, &q, true)) {
        e2fgvi_set_error("packed_conv_weight_f32x_size: bad geometry (channels per source must be multiples of 4)");
        return E2FGVI_EINVAL;
    }
    return q.total;
}

extern "C" int e2fgvi_pack_conv_weight_f32x(const float* w, float* wpacked, int32_t Cout, int32_t groups, int32_t KH,
                                            int32_t KW, int32_t nsrc, const int32_t* src_cpg, void* stream) {
    PackX q;
    E2_REQUIRE(w && wpacked && src_cpg, E2FGVI_EINVAL, "pack_conv_weight_f32x: null pointer");
    E2_REQUIRE(geometry_x(Cout, groups, KH, KW, nsrc, src_cpg, &q, true), E2FGVI_EINVAL, "pack_conv_weight_f32x: bad geometry");
    hipLaunchKernelGGL(pack_conv_weight_x_kernel<float>, dim3((unsigned)cdiv64(q.total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, wpacked, q);
    E2_LAUNCH_CHECK("pack_conv_weight_f32x");
    return 0;
}

/* X3 (fp32 on the bf16 matrix pipe): the fp32 geometry, three bf16 planes per weight -> 3 x the element count, in bf16 */
extern "C" int64_t e2fgvi_packed_conv_weight_f32x3_size(int32_t Cout, int32_t groups, int32_t KH, int32_t KW, int32_t nsrc,
                                                        const int32_t* src_cpg) {
    PackX q;
    if (!src_cpg || !geometry_x(Cout, groups, KH, KW, nsrc, src_cpg, &q, true)) {
        e2fgvi_set_error("packed_conv_weight_f32x3_size: bad geometry (channels per source must be multiples of 4)");
        return E2FGVI_EINVAL;
    }
    return 3 * q.total;
}

extern "C" int e2fgvi_pack_conv_weight_f32x3(const float* w, void* wpacked, int32_t Cout, int32_t groups, int32_t KH,
                                             int32_t KW, int32_t nsrc, const int32_t* src_cpg, void* stream) {
    PackX q;
    E2_REQUIRE(w && wpacked && src_cpg, E2FGVI_EINVAL, "pack_conv_weight_f32x3: null pointer");
    E2_REQUIRE(geometry_x(Cout, groups, KH, KW, nsrc, src_cpg, &q, true), E2FGVI_EINVAL, "pack_conv_weight_f32x3: bad geometry");
    hipLaunchKernelGGL(pack_conv_weight_x3_kernel, dim3((unsigned)cdiv64(q.total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, (unsigned short*)wpacked, q);
    E2_LAUNCH_CHECK("pack_conv_weight_f32x3");
    return 0;
}

// tap-packed variant of the packing: one source of fewer channels than two K-steps (a multiple of the chunk), no groups, KW >= 2
static bool geometry_taps(int Cout, int KH, int KW, int cin, PackX* q, bool f32 = false) {
    const int32_t cpg[1] = {cin};
    if (cin > 56 || KW < 2 || !geometry_x(Cout, 1, KH, KW, 1, cpg, q, f32)) return false;
    q->tp_cq = cin / q->ch;
    const int steps = cdiv(KH * KW * q->tp_cq, 8);
    q->wgroup_elems = (long long)steps * q->kc * q->Npad;
    q->total = q->wgroup_elems;
    return true;
}

extern "C" int64_t e2fgvi_packed_conv_weight_bf16x_taps_size(int32_t Cout, int32_t KH, int32_t KW, int32_t cin) {
    PackX q;
    if (!geometry_taps(Cout, KH, KW, cin, &q)) {
        e2fgvi_set_error("packed_conv_weight_bf16x_taps_size: one source of 8 ... 56 channels (multiple of 8), KW >= 2");
        return E2FGVI_EINVAL;
    }
    return q.total;
}

extern "C" int e2fgvi_pack_conv_weight_bf16x_taps(const float* w, void* wpacked, int32_t Cout, int32_t KH, int32_t KW, int32_t cin,
                                                  void* stream) {
    PackX q;
    E2_REQUIRE(w && wpacked, E2FGVI_EINVAL, "pack_conv_weight_bf16x_taps: null pointer");
    E2_REQUIRE(geometry_taps(Cout, KH, KW, cin, &q), E2FGVI_EINVAL, "pack_conv_weight_bf16x_taps: bad geometry");
    hipLaunchKernelGGL(pack_conv_weight_x_kernel<__bf16>, dim3((unsigned)cdiv64(q.total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, (__bf16*)wpacked, q);
    E2_LAUNCH_CHECK("pack_conv_weight_bf16x_taps");
    return 0;
}

extern "C" int64_t e2fgvi_packed_conv_weight_f32x_taps_size(int32_t Cout, int32_t KH, int32_t KW, int32_t cin) {
    PackX q;
    if (!geometry_taps(Cout, KH, KW, cin, &q, true)) {
        e2fgvi_set_error("packed_conv_weight_f32x_taps_size: one source of 4 ... 56 channels (multiple of 4), KW >= 2");
        return E2FGVI_EINVAL;
    }
    return q.total;
}

extern "C" int e2fgvi_pack_conv_weight_f32x_taps(const float* w, float* wpacked, int32_t Cout, int32_t KH, int32_t KW, int32_t cin,
                                                 void* stream) {
    PackX q;
    E2_REQUIRE(w && wpacked, E2FGVI_EINVAL, "pack_conv_weight_f32x_taps: null pointer");
    E2_REQUIRE(geometry_taps(Cout, KH, KW, cin, &q, true), E2FGVI_EINVAL, "pack_conv_weight_f32x_taps: bad geometry");
    hipLaunchKernelGGL(pack_conv_weight_x_kernel<float>, dim3((unsigned)cdiv64(q.total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, wpacked, q);
    E2_LAUNCH_CHECK("pack_conv_weight_f32x_taps");
    return 0;
}

extern "C" int64_t e2fgvi_packed_conv_weight_f32x3_taps_size(int32_t Cout, int32_t KH, int32_t KW, int32_t cin) {
    PackX q;
    if (!geometry_taps(Cout, KH, KW, cin, &q, true)) {
        e2fgvi_set_error("packed_conv_weight_f32x3_taps_size: one source of 4 ... 56 channels (multiple of 4), KW >= 2");
        return E2FGVI_EINVAL;
    }
    return 3 * q.total;
}

extern "C" int e2fgvi_pack_conv_weight_f32x3_taps(const float* w, void* wpacked, int32_t Cout, int32_t KH, int32_t KW, int32_t cin,
                                                  void* stream) {
    PackX q;
    E2_REQUIRE(w && wpacked, E2FGVI_EINVAL, "pack_conv_weight_f32x3_taps: null pointer");
    E2_REQUIRE(geometry_taps(Cout, KH, KW, cin, &q, true), E2FGVI_EINVAL, "pack_conv_weight_f32x3_taps: bad geometry");
    hipLaunchKernelGGL(pack_conv_weight_x3_kernel, dim3((unsigned)cdiv64(q.total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, (unsigned short*)wpacked, q);
    E2_LAUNCH_CHECK("pack_conv_weight_f32x3_taps");
    return 0;
}

static int conv2d_x(const e2fgvi_convx_desc* d, void* stream, int mode) {
    const bool f32 = mode != 0;           // fp32 activations (MODE 1: fp32 weights too; MODE 2: three bf16 planes per weight)
    E2_REQUIRE(d, E2FGVI_EINVAL, "conv2d_bf16x: null descriptor");
    PackX q;
    const int esz = f32 ? 4 : 2;
    const int wbytes_num = mode == 2 ? 6 : esz;       // packed bytes per weight
    E2_REQUIRE(geometry_x(d->Cout, d->groups, d->KH, d->KW, d->nsrc, d->src_cpg, &q, f32), E2FGVI_EINVAL,
               "conv2d_bf16x: bad geometry (channels per source must be multiples of 8 bf16 / 4 fp32)");
    E2_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Ho > 0 && d->Wo > 0 && d->stride > 0 && d->pad >= 0, E2FGVI_EINVAL,
               "conv2d_bf16x: bad sizes");
    if (d->out_grid) {
        // explicit output grid: `pad` rows above / `pad_left` columns left of the image, whatever the kernel reaches below /
        // right of it reads as zeros; Ho x Wo as given (the phase convolutions of SoftComp: 2-tap kernels with the zero row
        // on one side only); optionally scattered into a larger image
        E2_REQUIRE(d->pad_left >= 0 && d->act != E2FGVI_ACT_DCNPOST && !d->dst_nchw && !d->tap_packed && (d->tile < 10 || d->tile > 20),
                   E2FGVI_EINVAL, "conv2d_bf16x: explicit output grids take plain NHWC tiles only");
        if (d->out_sy || d->out_sx)
            E2_REQUIRE(d->out_sy > 0 && d->out_sx > 0 && d->out_py >= 0 && d->out_px >= 0 && d->out_H > 0 && d->out_W > 0 &&
                       (d->Ho - 1) * d->out_sy + d->out_py < d->out_H && (d->Wo - 1) * d->out_sx + d->out_px < d->out_W &&
                       (long long)d->N * d->out_H * d->out_W < 2147483647LL,
                       E2FGVI_EINVAL, "conv2d_bf16x: the output scatter leaves the [N, out_H, out_W] image");
    } else {
        E2_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1 && d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1,
                   E2FGVI_EINVAL, "conv2d_bf16x: Ho/Wo inconsistent with H/W/k/stride/pad");
        E2_REQUIRE(!d->out_sy && !d->out_sx && !d->pad_left, E2FGVI_EINVAL, "conv2d_bf16x: pad_left / out_s* need out_grid = 1");
    }
    E2_REQUIRE(!d->res_bcast || (d->residual && d->act != E2FGVI_ACT_DCNPOST), E2FGVI_EINVAL, "conv2d_bf16x: res_bcast without a residual");
    E2_REQUIRE((long long)d->N * d->Ho * d->Wo < 2147483647LL, E2FGVI_EUNSUP, "conv2d_bf16x: more than 2^31 output pixels");
    E2_REQUIRE(d->KH <= 8 && d->KW <= 8, E2FGVI_EUNSUP, "conv2d_bf16x: kernels larger than 8x8 are not supported");
    E2_REQUIRE(d->wpacked && d->dst, E2FGVI_EINVAL, "conv2d_bf16x: null weight/dst");
    E2_REQUIRE((d->dst_dtype == E2FGVI_F32 || d->dst_dtype == E2FGVI_BF16) && (d->res_dtype == E2FGVI_F32 || d->res_dtype == E2FGVI_BF16),
               E2FGVI_EINVAL, "conv2d_bf16x: dtype must be E2FGVI_F32 or E2FGVI_BF16");
    ConvXParams p;
    for (int s = 0; s < E2FGVI_MAX_SRC; ++s) { p.src[s] = nullptr; p.ld[s] = 0; p.coff[s] = 0; p.cpg[s] = 0; p.src_bytes[s] = 0; }
    for (int s = 0; s < d->nsrc; ++s) {
        E2_REQUIRE(d->src[s], E2FGVI_EINVAL, "conv2d_bf16x: null source %d", s);
        E2_REQUIRE(d->src_ld[s] % q.ch == 0 && d->src_coff[s] % q.ch == 0 && ((uintptr_t)d->src[s] & 15) == 0, E2FGVI_EINVAL,
                   "conv2d_bf16x: source %d not 16-byte addressable (ld / coff multiples of 8 bf16 / 4 fp32)", s);
        E2_REQUIRE(d->src_coff[s] + d->groups * d->src_cpg[s] <= d->src_ld[s], E2FGVI_EINVAL,
                   "conv2d_bf16x: source %d channel range exceeds its pixel stride", s);
        const long long bytes = (long long)d->N * d->H * d->W * d->src_ld[s] * esz;
        E2_REQUIRE(bytes < 4294967295LL, E2FGVI_EUNSUP, "conv2d_bf16x: source %d spans >= 4 GiB (split the batch)", s);
        p.src[s] = d->src[s]; p.ld[s] = d->src_ld[s]; p.coff[s] = d->src_coff[s]; p.cpg[s] = d->src_cpg[s];
        p.src_bytes[s] = (unsigned)bytes;
    }
    E2_REQUIRE(q.wgroup_elems * wbytes_num < 4294967295LL, E2FGVI_EUNSUP, "conv2d_bf16x: packed weight group >= 4 GiB");
    E2_REQUIRE(((uintptr_t)d->wpacked & 15) == 0, E2FGVI_EINVAL, "conv2d_bf16x: packed weight not 16-byte aligned");
    if (d->dst_nchw)
        E2_REQUIRE(d->dst_dtype == E2FGVI_F32 && !d->dst2, E2FGVI_EINVAL, "conv2d_bf16x: the NCHW destination is fp32, without a second copy");
    else
        E2_REQUIRE(d->dst_coff >= 0 && d->dst_coff + d->Cout <= d->dst_ld, E2FGVI_EINVAL, "conv2d_bf16x: dst slice exceeds dst_ld");
    E2_REQUIRE(((uintptr_t)d->dst & 15) == 0 && (!d->dst2 || ((uintptr_t)d->dst2 & 15) == 0) &&
               (!d->residual || ((uintptr_t)d->residual & 15) == 0), E2FGVI_EINVAL, "conv2d_bf16x: dst / dst2 / residual not 16-byte aligned");
    if (d->dst2_plane_stride) {
        // ABI 8: channels from dst2_split_from on as three exact bf16 planes (16-byte stores of 8 channels)
        E2_REQUIRE(d->dst2 && d->dst2_plane_stride > 0 && d->dst_dtype == E2FGVI_F32 && !d->dst_nchw && d->groups == 1 &&
                       !(d->out_grid && d->out_sy), E2FGVI_EINVAL,
                   "conv2d_bf16x: split planes need dst2, an fp32 NHWC dst, groups = 1 and no output scatter");
        E2_REQUIRE(d->Cout % 8 == 0 && d->dst2_split_from >= 0 && d->dst2_split_from <= d->Cout && d->dst2_split_from % 8 == 0 &&
                       d->dst2_ld % 8 == 0 && d->dst2_coff % 8 == 0 && d->dst2_plane_stride % 8 == 0 && d->dst2_coff >= 0 &&
                       d->dst2_coff + d->Cout - d->dst2_split_from <= d->dst2_ld, E2FGVI_EINVAL,
                   "conv2d_bf16x: split planes: Cout, dst2_split_from, dst2_ld, dst2_coff and dst2_plane_stride in multiples of 8, "
                   "the split channels inside dst2_ld");
    } else if (d->dst2)
        E2_REQUIRE(d->dst2_coff >= 0 && d->dst2_coff + d->Cout <= d->dst2_ld, E2FGVI_EINVAL, "conv2d_bf16x: dst2 slice exceeds dst2_ld");
    if (d->act == E2FGVI_ACT_DCNPOST)
        E2_REQUIRE(d->residual && d->res_dtype == E2FGVI_F32 && d->Cout % 3 == 0 && d->groups == 1, E2FGVI_EINVAL,
                   "conv2d_bf16x: ACT_DCNPOST needs the fp32 [pixel][4] flows as residual, Cout %% 3 == 0, groups == 1");
    p.nsrc = d->nsrc;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Ho = d->Ho; p.Wo = d->Wo;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
    p.padx = d->out_grid ? d->pad_left : d->pad;
    p.osy = d->out_grid ? d->out_sy : 0; p.osx = d->out_sx; p.opy = d->out_py; p.opx = d->out_px; p.oH = d->out_H; p.oW = d->out_W;
    p.res_bcast = d->res_bcast;
    p.Cout = d->Cout; p.Cout_g = q.Cout_g; p.Npad = q.Npad;
    p.M = d->N * d->Ho * d->Wo;
    p.nsteps = d->KH * d->KW * q.steps_per_tap;
    p.tp_cq = 0; p.tp_magic = 0; p.tp_magic_cq = 0;
    if (d->tap_packed) {
        PackX qt;
        E2_REQUIRE(d->nsrc == 1 && d->groups == 1 && geometry_taps(d->Cout, d->KH, d->KW, d->src_cpg[0], &qt, f32), E2FGVI_EINVAL,
                   "conv2d_bf16x: tap-packed weights are for one source of <= 56 channels, no groups, KW >= 2");
        E2_REQUIRE(d->tile < 10 || d->tile > 20, E2FGVI_EUNSUP, "conv2d_bf16x: the row-shift tiles do not take tap-packed weights");
        q.wgroup_elems = qt.wgroup_elems;
        p.tp_cq = qt.tp_cq;
        p.tp_magic = 0xFFFFFFFFu / (unsigned)d->KW + 1u;
        p.tp_magic_cq = qt.tp_cq > 1 ? 0xFFFFFFFFu / (unsigned)qt.tp_cq + 1u : 0u;
        p.nsteps = cdiv(d->KH * d->KW * qt.tp_cq, 8);
    }
    p.wgroup_elems = q.wgroup_elems; p.wgroup_bytes = (unsigned)(q.wgroup_elems * wbytes_num);
    p.w = d->wpacked; p.bias = d->bias;
    p.res = d->residual; p.res_ld = d->res_ld; p.res_coff = d->res_coff; p.res_bf16 = d->res_dtype == E2FGVI_BF16;
    p.dst = d->dst; p.dst_ld = d->dst_ld; p.dst_coff = d->dst_coff; p.dst_bf16 = d->dst_dtype == E2FGVI_BF16;
    p.dst_nchw = d->dst_nchw;
    p.dst2 = (__bf16*)d->dst2; p.dst2_ld = d->dst2_ld; p.dst2_coff = d->dst2_coff;
    p.split_from = d->dst2_split_from; p.split_plane = d->dst2_plane_stride;
    p.act = d->act; p.slope = d->slope;
    hipStream_t st = (hipStream_t)stream;
    int tile = d->tile;
    p.dbg_noload = 0;
    if (tile == 21 || tile == 24 || tile == 26 || tile == 27) { p.dbg_noload = 1; tile -= 20; }        // measurement aids: results are garbage
    if (tile == 31 || tile == 34 || tile == 36) { p.dbg_noload = 2; tile -= 30; }        //   no epilogue
    if (tile == 41 || tile == 44 || tile == 46) { p.dbg_noload = 3; tile -= 40; }        //   neither
    if (tile == 61 || tile == 66) { p.dbg_noload = 8; tile -= 60; }        //   epilogue without its global stores
    if (tile == 71 || tile == 74 || tile == 76 || tile == 77) { p.dbg_noload = 16; tile -= 70; }   //   no A (activation) DMA
    if (tile == 81 || tile == 84 || tile == 86 || tile == 87) { p.dbg_noload = 32; tile -= 80; }   //   no B (weight) DMA
    if (!tile) {
        if (p.Cout_g <= 32) tile = 3;
        else if (p.Cout_g <= 64) tile = 2;
        else tile = ((long long)cdiv(p.M, 128) * cdiv(p.Cout_g, 128) * d->groups >= 384) ? 1 : 4;
    }
    switch (tile) {
        case 1: return launch_x<128, 128, 2, 2, false>(p, d->groups, st, mode);
        case 2: return launch_x<128, 64, 2, 2, false>(p, d->groups, st, mode);
        case 3: return launch_x<128, 32, 4, 1, false>(p, d->groups, st, mode);
        case 4: return launch_x<64, 128, 2, 2, false>(p, d->groups, st, mode);
        case 5: return launch_x<64, 64, 2, 2, false>(p, d->groups, st, mode);
        case 6: return launch_x<256, 128, 4, 2, false>(p, d->groups, st, mode);
        case 7: return launch_x<256, 256, 4, 2, false>(p, d->groups, st, mode);     // 8 waves x (64 x 128): half the DMA per FLOP of tile 1
        case 8: return launch_x<256, 192, 4, 2, false>(p, d->groups, st, mode);     // 8 waves x (64 x 96): N = 1536 in 8 column tiles (qkv: 232 instead of 174 workgroups)
        // split-operand mode with the two halves of the workgroup half a step apart (kernel comment, PP)
        case 107: return launch_x<256, 256, 4, 2, false, true>(p, d->groups, st, mode);
        case 108: return launch_x<256, 192, 4, 2, false, true>(p, d->groups, st, mode);
        // 3x3 stride-1 pad-1 layers: the three horizontal taps share one A stage (row-shifted reads)
        case 11: return launch_x<128, 128, 2, 2, true>(p, d->groups, st, mode);
        case 12: return launch_x<128, 64, 2, 2, true>(p, d->groups, st, mode);
        case 13: return launch_x<128, 32, 4, 1, true>(p, d->groups, st, mode);      // narrow layers (decoder tail): the A stream is
        case 14: return launch_x<64, 128, 2, 2, true>(p, d->groups, st, mode);
        case 16: return launch_x<256, 128, 4, 2, true>(p, d->groups, st, mode);
        case 17: return launch_x<256, 256, 4, 2, true>(p, d->groups, st, mode);
        case 18: return launch_x<256, 64, 4, 2, true>(p, d->groups, st, mode);       // all of their traffic, a third of it here
        default: break;
    }
    e2fgvi_set_error("conv2d_bf16x: unknown tile %d", tile);
    return E2FGVI_EINVAL;
}

extern "C" int e2fgvi_conv2d_bf16x(const e2fgvi_convx_desc* d, void* stream) { return conv2d_x(d, stream, 0); }
/* the same kernel on fp32 operands (fp32 NHWC sources, e2fgvi_pack_conv_weight_f32x weights, exact fp32 MFMA) */
extern "C" int e2fgvi_conv2d_f32x(const e2fgvi_convx_desc* d, void* stream) { return conv2d_x(d, stream, 1); }
/* fp32 NHWC sources, e2fgvi_pack_conv_weight_f32x3 weights: fp32 products as six exact bf16 MFMA terms (kernel MODE 2) */
extern "C" int e2fgvi_conv2d_f32x3(const e2fgvi_convx_desc* d, void* stream) { return conv2d_x(d, stream, 2); }
