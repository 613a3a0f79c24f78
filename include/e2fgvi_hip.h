/* e2fgvi_hip.h -- C ABI of libe2fgvi_hip.so (MI355X / gfx950 kernels for the E2FGVI forward).
 *
 * The reference (MCG-NKU/E2FGVI) is pure Python: every native instruction it executes is reached
 * through torch / mmcv operator calls.  This header is the replacement for that operator boundary
 * on the inference hot path: one entry point per fused stage.  Each entry
 * cites the reference operator call it replaces (file:line under /root/reference).
 *
 * Conventions
 *   - raw device pointers (fp32 unless stated), explicit int dims, caller-owned buffers: no
 *     allocation, no synchronisation, no host<->device copies inside;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - activations are NHWC ("channels last"): element (n,y,x,c) at ((n*H+y)*W+x)*ld + c, where the
 *     pixel stride `ld` (in floats) may exceed C so that callers can address channel slices;
 *   - returns 0 on success, a negative E2FGVI_E* code on bad arguments, a positive hipError_t on a
 *     launch failure; e2fgvi_last_error() gives the thread-local message;
 *   - re-entrant; no global mutable state.
 */
#ifndef E2FGVI_HIP_H
#define E2FGVI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define E2FGVI_EINVAL (-1)   /* bad argument */
#define E2FGVI_EUNSUP (-2)   /* valid but unsupported configuration */

#define E2FGVI_ACT_NONE 0
#define E2FGVI_ACT_RELU 1
#define E2FGVI_ACT_LRELU 2   /* slope in desc */
#define E2FGVI_ACT_TANH 3
/* ------------------------------------------------------------------------------------------------
 * Byte side of the sliding-window video driver (reference: test.py).
 * uint8 frames [L,H,W,3] and masks stay on the device; all results are bit-exact with the numpy / PIL reference.
 * ------------------------------------------------------------------------------------------------ */
/* test.py:56-69 read_mask: NEAREST resize of [L,Hin,Win] masks to H x W (ytab[H] / xtab[W] = source row / column of
 * every output row / column, built like Pillow's ImagingScaleAffine), binarise (> 0), `iterations` dilations with the
 * 3x3 cross (cv2.dilate, out-of-image pixels ignored).  out: [L,H,W] of 0 / 1. */
int e2fgvi_mask_prepare(const uint8_t* masks, int32_t L, int32_t Hin, int32_t Win, const int32_t* ytab, const int32_t* xtab,
                        uint8_t* out, int32_t H, int32_t W, int32_t iterations, void* stream);
/* test.py:146-165: clip[ti][c][y][x] = (frames[ids[ti]]/255*2-1) * (1 - masks[ids[ti]]), fp32 NCHW [t,3,Hp,Wp], rows /
 * columns beyond H / W mirror the frame (cat([x, flip(x)])[:Hp]). */
int e2fgvi_masked_clip(const uint8_t* frames, const uint8_t* masks, const int32_t* ids, int32_t t, int32_t H, int32_t W,
                       float* clip, int32_t Hp, int32_t Wp, void* stream);
/* test.py:168-179: for the first n frames of pred [*,3,Hp,Wp] (model output in (-1,1)):
 * img = uint8((pred+1)/2*255) * mask + frame * (1-mask); comp[ids[i]] = first[i] ? img : comp*0.5 + img*0.5 (float [L,H,W,3]). */
int e2fgvi_composite(const float* pred, const int32_t* ids, const uint8_t* first, int32_t n, const uint8_t* frames,
                     const uint8_t* masks, float* comp, int32_t H, int32_t W, int32_t Hp, int32_t Wp, void* stream);
/* ndarray.astype(uint8) of the blended frames (truncation) */
int e2fgvi_float_to_u8(const float* src, uint8_t* dst, int64_t n, void* stream);
/* model output [N,3,Hp,Wp] in (-1,1) -> uint8 NHWC [N,H,W,3] = uint8((pred+1)/2*255): the form the clip-sharded runner
 * gathers over xGMI (4x fewer bytes than fp32). */
int e2fgvi_pred_to_u8(const float* pred, uint8_t* dst, int32_t N, int32_t H, int32_t W, int32_t Hp, int32_t Wp, void* stream);

/* ------------------------------------------------------------------------------------------------
 * evaluate.py metrics (core/metrics.py:20-56): per image pair of fp32 NHWC [N,H,W,3] frames in
 * [0,255]: out[2n] = PSNR (inf when identical), out[2n+1] = SSIM as skimage compare_ssim(data_range=255,
 * multichannel=True, win_size) computes it (uniform window, sample covariance, K1 .01, K2 .03, crop (win-1)/2).
 * fp64 like the reference.  workspace: e2fgvi_psnr_ssim_workspace(N,H,W) bytes, caller-owned.
 * ------------------------------------------------------------------------------------------------ */
int64_t e2fgvi_psnr_ssim_workspace(int32_t N, int32_t H, int32_t W);
int e2fgvi_psnr_ssim(const float* img1, const float* img2, int32_t N, int32_t H, int32_t W, int32_t win_size,
                     void* workspace, double* out, void* stream);

/* conv_offset post-processing of SecondOrderDeformableAlignment (feat_prop.py:38-53) fused into the conv that produces
 * it: `residual` must point to the per-pixel flows [P,4] = (u1,v1,u2,v2) with res_ld = 4, `slope` = max_residue:
 * offset channels -> slope*tanh(v) + flow.flip, mask channels (last third) -> sigmoid(v) */
#define E2FGVI_ACT_DCNPOST 4

#define E2FGVI_MAX_SRC 4

/* element types of the tensors of the bf16 data path (see the end of this header) */
#define E2FGVI_F32 0
#define E2FGVI_BF16 1
#define E2FGVI_BF16X3 2      /* mdcn mfma_dtype (ABI 7): fp32 sources and results, the MFMA operands as three bf16 pieces each, six exact terms per product */

const char* e2fgvi_last_error(void);
int e2fgvi_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear layer on fp32 MFMA (v_mfma_f32_32x32x2_f32).
 * Replaces torch.nn.functional.conv2d / linear at: encoder model/e2fgvi.py:75-109, decoder
 * :143-150, SPyNet 7x7 stacks model/modules/flow_comp.py:180-215, conv_offset / backbone / fusion
 * model/modules/feat_prop.py:20-28,73-79, SoftSplit unfold+linear tfocal_transformer.py:40-45
 * (= 7x7 stride-3 conv), qkv/proj :221,398, FFN linears :80-81, SoftComp linear :68.
 *
 * The input is the *virtual channel concat* of up to 4 NHWC sources (so torch.cat copies at
 * e2fgvi.py:101-107, feat_prop.py:36,126-136 never exist).  With `groups` > 1, group g reads
 * channels [coff[s] + g*cpg[s], +cpg[s]) of every source s, in source order.
 * Weights must be pre-packed with e2fgvi_pack_conv_weight using the same cpg[] / bk.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const float* src[E2FGVI_MAX_SRC];
    int32_t src_ld[E2FGVI_MAX_SRC];    /* pixel stride of each source, floats (multiple of 4)      */
    int32_t src_coff[E2FGVI_MAX_SRC];  /* first channel used by group 0 (multiple of 4)            */
    int32_t src_cpg[E2FGVI_MAX_SRC];   /* channels per group taken from this source (mult. of 4)   */
    int32_t nsrc;
    int32_t N, H, W;                   /* input batch / height / width                             */
    int32_t Ho, Wo;                    /* output height / width                                    */
    int32_t KH, KW, stride, pad;
    int32_t groups;
    int32_t Cout;                      /* total output channels                                    */
    int32_t bk;                        /* K-chunk the weights were packed for: 8, 16 or 32         */
    const float* wpacked;
    const float* bias;                 /* [Cout] or NULL                                           */
    const float* residual;             /* NHWC [N,Ho,Wo,*] added before the activation, or NULL    */
    int32_t res_ld, res_coff;
    float* dst;
    int32_t dst_ld, dst_coff;
    int32_t dst_nchw;                  /* 1: dst is plain NCHW [N,Cout,Ho,Wo] (ld/coff ignored)     */
    int32_t act;
    float slope;
    int32_t tile;                      /* 0 = auto; otherwise force a tile config (tests/bench)    */
} e2fgvi_conv_desc;

int e2fgvi_conv2d_nhwc(const e2fgvi_conv_desc* d, void* stream);
/* The same operator from a build without packed-fp32 VALU instructions (v_pk_{mul,add,fma}_f32): for launches that run
 * on a side stream concurrently with bf16 MFMA kernels -- SPyNet next to the encoder.  Identical results. */
int e2fgvi_conv2d_nhwc_nopk(const e2fgvi_conv_desc* d, void* stream);

/* number of floats of the packed weight buffer for the given geometry */
int64_t e2fgvi_packed_conv_weight_size(int32_t Cout, int32_t groups, int32_t KH, int32_t KW,
                                       int32_t nsrc, const int32_t* src_cpg, int32_t bk);
/* w: reference layout [Cout, sum(cpg), KH, KW] (torch OIHW; Linear = [Cout, Cin, 1, 1]) */
int e2fgvi_pack_conv_weight(const float* w, float* wpacked, int32_t Cout, int32_t groups,
                            int32_t KH, int32_t KW, int32_t nsrc, const int32_t* src_cpg,
                            int32_t bk, void* stream);

/* Winograd F(2x2,3x3) form of the same operator for 3x3 / stride 1 / pad 1 layers with even H, W (the encoder's
 * stride-1 layers e2fgvi.py:77-93, the decoder convs :112-150, SoftComp's HQ bias conv e2fgvi_hq tfocal :67-79): fp32
 * arithmetic on the fp32 MFMA pipe, 16 instead of 36 multiplies per 2x2 outputs.  Also the propagation convs
 * (feat_prop.py:20-28,73-79) incl. the ACT_DCNPOST epilogue.  Same descriptor; restrictions: every src_cpg a multiple
 * of 4, NHWC output, bk ignored; tile = 0 (auto), 32 / 64 (couts per workgroup, 16x16-pixel blocks) or 132 / 164
 * (8x16-pixel blocks).  Weights: [group][8-channel chunk][16 positions][2][Npad][4] holding G g G^T. */
int64_t e2fgvi_packed_winograd_weight_size(int32_t Cout, int32_t groups, int32_t nsrc, const int32_t* src_cpg); /* floats */
int e2fgvi_pack_winograd_weight(const float* w, float* wpacked, int32_t Cout, int32_t groups, int32_t nsrc,
                                const int32_t* src_cpg, void* stream);
int e2fgvi_conv3x3_winograd(const e2fgvi_conv_desc* d, void* stream);

/* Wide-tile Winograd forms of the same operator (csrc/conv_wino4.hip): F(fy x 4, 3x3) with fy = 2 (24 transform positions
 * per 2x4 outputs: 3 multiplies per output and input channel) or fy = 4 (36 per 4x4: 2.25), against 4 for F(2x2,3x3) and
 * 9 for the direct convolution the reference runs (F.conv2d at e2fgvi.py:77-93,112-150, feat_prop.py:20-28,73-79).
 * Same descriptor and restrictions as e2fgvi_conv3x3_winograd, plus H % fy == 0 and W % 4 == 0; tile = 0 (auto), 32 or
 * 64 couts per workgroup (fy = 4: 32).  Weights: [group][8-channel chunk][(fy+2)*6 positions][2][Npad][4] holding
 * G_y g G_x^T.  fp32 throughout; rounding relative to the output rms on 512 input channels: 6.5e-6 (fy = 2), 1.9e-5
 * (fy = 4) -- the direct fp32 convolution sits at 7.7e-6. */
int64_t e2fgvi_packed_winograd4_weight_size(int32_t Cout, int32_t groups, int32_t nsrc, const int32_t* src_cpg, int32_t fy);
int e2fgvi_pack_winograd4_weight(const float* w, float* wpacked, int32_t Cout, int32_t groups, int32_t nsrc,
                                 const int32_t* src_cpg, int32_t fy, void* stream);
int e2fgvi_conv3x3_winograd4(const e2fgvi_conv_desc* d, int32_t fy, void* stream);

/* The decoder's last layer (csrc/conv_tail.hip): nn.Conv2d(64, 3, kernel_size=3, stride=1, padding=1) + torch.tanh
 * (model/e2fgvi.py:99-103,261 / model/e2fgvi_hq.py:99-103,263).  With 3 output channels the nine taps move to the N side of
 * ONE [pixels x 64] x [64 x 27] GEMM (every input pixel read once, 9x fewer matrix instructions than the implicit GEMM),
 * followed by the shifted 9-term sum in LDS.  src: NHWC [N,H,W,src_ld >= 64] of src_dtype (E2FGVI_F32: exact fp32 MFMA;
 * E2FGVI_BF16: bf16 MFMA, fp32 accumulation), 16-byte aligned rows; wpacked: 64 x 32 elements of src_dtype from
 * e2fgvi_pack_tail_weight (w: fp32 OIHW [3,64,3,3]); bias fp32 [3] or NULL; dst fp32 NCHW [N,3,H,W]; act: E2FGVI_ACT_*.
 * Only Cin = 64, Cout = 3 is built (E2FGVI_EUNSUP otherwise). */
int64_t e2fgvi_packed_tail_weight_size(int32_t Cout, int32_t Cin);
int e2fgvi_pack_tail_weight(const float* w, void* wpacked, int32_t Cout, int32_t Cin, int32_t dtype, void* stream);
int e2fgvi_conv3x3_tail(const void* src, int32_t src_dtype, int32_t src_ld, const void* wpacked, const float* bias, float* dst,
                        int32_t N, int32_t H, int32_t W, int32_t act, float slope, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Modulated deformable convolution (DCNv2), im2col-free: bilinear gather straight into LDS + MFMA.
 * Replaces mmcv.ops.modulated_deform_conv2d (mmcv-full 1.4.8) called at
 * model/modules/feat_prop.py:55-58.  x is the virtual concat of two NHWC sources (feat_prop |
 * feat_n2, feat_prop.py:127).  offset: [P, dg*2*K] with (dy,dx) interleaved per tap, mask:
 * [P, dg*K] (both pixel-major, the NHWC image of mmcv's NCHW tensors).
 * If `flows` != NULL the kernel also applies SecondOrderDeformableAlignment's post-processing
 * (feat_prop.py:38-53) on the fly: offset/mask are then the raw conv_offset output (o1|o2|mask),
 * offset = max_residue * tanh(raw) + flow.flip (flow_1 for the first dg/2 groups, flow_2 for the
 * rest), mask = sigmoid(raw); flows is [P,4] = (u1,v1,u2,v2).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const void* src[2];                /* fp32 NHWC (bf16 NHWC with src_dtype = E2FGVI_BF16)       */
    int32_t src_ld[2];
    int32_t src_c[2];                  /* channels of each source; C = sum; C/dg multiple of 16    */
    int32_t nsrc;
    int32_t N, H, W, Ho, Wo;
    int32_t KH, KW, stride, pad, dil;
    int32_t deform_groups;
    int32_t Cout;
    const float* offset; int32_t off_ld;
    const float* mask;   int32_t mask_ld;
    const float* flows;                /* optional [P,4]                                           */
    float max_residue;
    const float* wpacked;              /* e2fgvi_pack_dcn_weight                                   */
    const float* bias;
    float* dst; int32_t dst_ld, dst_coff;
    int32_t tile;
    int32_t dst_dtype;                 /* E2FGVI_F32 (0, default) or E2FGVI_BF16: dst is a bf16 NHWC tensor        */
    int32_t mfma_dtype;                /* E2FGVI_F32 (0, default): fp32 MFMA, wpacked from e2fgvi_pack_dcn_weight;
                                          E2FGVI_BF16: the sampled slab is rounded to bf16 and multiplied on bf16 MFMA
                                          (fp32 gather / blend / accumulation), wpacked from e2fgvi_pack_dcn_weight_bf16 */
    int32_t src_dtype;                 /* E2FGVI_F32 (0, default); E2FGVI_BF16 (needs mfma_dtype = E2FGVI_BF16): the sources
                                          are bf16 NHWC (src_ld multiple of 8): half the gather fetches                 */
    int32_t src_planar;                /* 1 (ABI version 4; bf16 sources, 16 channels per deform group): source s is laid out
                                          [src_c[s] / 16 groups][N*H*W pixels][16 channels] (e2fgvi_nhwc_to_planar16) instead
                                          of NHWC -- the 32-byte runs of neighbouring pixels of one group are then adjacent in
                                          memory, so the corner fetches of neighbouring output pixels share cache lines       */
} e2fgvi_mdcn_desc;

int e2fgvi_mdcn_nhwc(const e2fgvi_mdcn_desc* d, void* stream);
int64_t e2fgvi_packed_dcn_weight_size(int32_t Cout, int32_t C, int32_t KH, int32_t KW);
/* w: [Cout, C, KH, KW] */
int e2fgvi_pack_dcn_weight(const float* w, float* wpacked, int32_t Cout, int32_t C, int32_t KH,
                           int32_t KW, int32_t deform_groups, void* stream);
/* split-operand packing (3 x e2fgvi_packed_dcn_weight_size elements of 2 bytes: three bf16 planes whose sum is the fp32 weight)
 * for mfma_dtype = E2FGVI_BF16X3 -- the fp32 deformable conv (feat_prop.py:55-58) on the bf16 matrix pipe, fp32-level rounding */
int e2fgvi_pack_dcn_weight_x3(const float* w, void* wpacked, int32_t Cout, int32_t C, int32_t KH, int32_t KW,
                              int32_t deform_groups, void* stream);
/* bf16 packing (e2fgvi_packed_dcn_weight_size elements of 2 bytes) for mfma_dtype = E2FGVI_BF16 */
int e2fgvi_pack_dcn_weight_bf16(const float* w, void* wpacked, int32_t Cout, int32_t C, int32_t KH,
                                int32_t KW, int32_t deform_groups, void* stream);

/* bf16 NHWC [P pixels][C] (C a multiple of 16) -> [C / 16][P][16]: the deformable conv's planar source layout (src_planar) */
int e2fgvi_nhwc_to_planar16(const void* src, void* dst, int64_t P, int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Temporal focal window attention, fused (flash-style, fp32 MFMA, online softmax).
 * Replaces WindowAttention.forward's roll/partition/cat/bmm/softmax/bmm chain,
 * model/modules/tfocal_transformer.py:226-396 (window 5x9, 4 heads of 128).
 *   qkv    [B*T*fh*fw, 1536]  rows in (b,t,y,x) order, columns q|k|v (tfocal_transformer.py:221-223)
 *   kv_pool[B*T*nWin, 1536]   qkv Linear applied to the pooled window tokens (:319), rows (b,t,win)
 *   key_tab[nWin, tab_ld]     per window: `nkeys[win]` key references per frame:
 *                             v >= 0 : token y*fw+x of the same frame (own window + rolled ring,
 *                                      duplicates included, :235-283);  v < 0 : pooled window -(v+1)
 *   nkeys  [nWin]             valid references per frame; the remaining (210 - nkeys) pooled slots
 *                             are the zero-padded ones that score exactly -100 (:301-316,378-380)
 *   out    [B*T*fh*fw, 512]   attention output in token order (window_reverse :132 is implicit)
 * ---------------------------------------------------------------------------------------------- */
int e2fgvi_focal_attention(const float* qkv, const float* kv_pool, const int32_t* key_tab,
                           int32_t tab_ld, const int32_t* nkeys, float* out, int32_t B, int32_t T,
                           int32_t fh, int32_t fw, int32_t waves, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Small HBM-bound kernels
 * ---------------------------------------------------------------------------------------------- */
/* [N,C,H,W] -> NHWC with pixel stride ld (channels C..ld-1 zero-filled); y = x*scale + shift */
int e2fgvi_nchw_to_nhwc(const float* src, float* dst, int32_t N, int32_t C, int32_t H, int32_t W,
                        int32_t ld, float scale, float shift, void* stream);
int e2fgvi_nhwc_to_nchw(const float* src, int32_t ld, float* dst, int32_t N, int32_t C, int32_t H,
                        int32_t W, void* stream);

/* Bilinear resize (torch F.interpolate semantics, align_corners 0/1), NHWC or NCHW source ->
 * NHWC destination, followed by a per-channel affine y = v*scale[c] + shift[c] (scale/shift may be
 * NULL).  Replaces e2fgvi.py:214-219 (1/4 downsample), flow_comp.py:150-167 (SPyNet resizes and
 * flow rescale), :121-124 (flow x2) and e2fgvi.py:126-129 (decoder x2). */
int e2fgvi_resize_bilinear(const float* src, int32_t src_nchw, int32_t src_ld, float* dst,
                           int32_t dst_ld, int32_t N, int32_t C, int32_t H, int32_t W, int32_t Ho,
                           int32_t Wo, int32_t align_corners, const float* scale,
                           const float* shift, void* stream);

/* 2x2 mean pooling, NHWC (flow_comp.py:101-111) */
int e2fgvi_avgpool2_nhwc(const float* src, float* dst, int32_t N, int32_t H, int32_t W, int32_t C,
                         void* stream);

/* SPyNet level input (flow_comp.py:117-132): for pair n, out[n,y,x,0:8] =
 * [ref(3), warp_border(supp, flow_up)(3), flow_up(2)], flow_up = 2 * up2x_align_corners(flow_prev)
 * (zeros when flow_prev == NULL).  pyr: [F,h,w,4] per-frame pyramid level; ref_idx/supp_idx: [Np]
 * frame indices; flow_prev: [Np,h/2,w/2,2]. */
int e2fgvi_spynet_level_input(const float* pyr, const int32_t* ref_idx, const int32_t* supp_idx,
                              const float* flow_prev, float* out, int32_t Np, int32_t h, int32_t w,
                              void* stream);

/* Propagation step conditions (feat_prop.py:110-123): given feat_prop, feat_n2 [N,H,W,C] and the
 * flow fields flow_a (=flows[:,i-1]) and flow_b (=flows[:,i-2] or NULL), all NHWC, writes
 *   cond  [N,H,W,2C] = warp0(feat_prop, flow_n1) | warp0(feat_n2, flow_n2)   (zeros padding)
 *   flows [N,H,W,4]  = flow_n1 | flow_n2,  flow_n2 = flow_n1 + warp0(flow_b, flow_n1) (0 if NULL)
 * Each of the N images has its own flow image: flow_x + n*flow_img_stride. */
int e2fgvi_prop_cond(const float* feat_prop, int32_t fp_ld, const float* feat_n2, int32_t f2_ld,
                     const float* flow_a, const float* flow_b, int64_t flow_img_stride,
                     float* cond, float* flows, int32_t N, int32_t H, int32_t W, int32_t C,
                     void* stream);

/* LayerNorm over the last dim (C multiple of 64, eps 1e-5, biased variance); tfocal_transformer.py
 * :452,463,470,533 */
int e2fgvi_layernorm(const float* x, const float* gamma, const float* beta, float* y, int64_t rows,
                     int32_t C, void* stream);

/* Window pooling Linear(45->1) (tfocal_transformer.py:508-516): x [B*T,fh,fw,C] tokens ->
 * pooled [B*T, fh/5, fw/9, C] */
int e2fgvi_window_pool(const float* x, const float* w45, const float* bias1, float* pooled,
                       int32_t BT, int32_t fh, int32_t fw, int32_t C, void* stream);

/* FusionFeedForward middle (tfocal_transformer.py:92-97): hid [F*fh*fw, C*49] ->
 * fold(7,3,3) / overlap count -> folded [F,H,W,C];  then unfold + exact GELU -> [F*fh*fw, C*49] */
int e2fgvi_ffn_fold(const float* hid, float* folded, int32_t F, int32_t fh, int32_t fw, int32_t H,
                    int32_t W, int32_t C, void* stream);
int e2fgvi_ffn_unfold_gelu(const float* folded, float* out, int32_t F, int32_t fh, int32_t fw,
                           int32_t H, int32_t W, int32_t C, void* stream);

/* SoftComp fold (tfocal_transformer.py:70-71): emb [F*fh*fw, C*49] -> overlap-ADD fold ->
 * [F,H,W,C] + bias_hwc[H,W,C] (optional) + residual (optional, NHWC ld = C) */
int e2fgvi_softcomp_fold(const float* emb, const float* bias_hwc, const float* residual,
                         float* dst, int32_t F, int32_t fh, int32_t fw, int32_t H, int32_t W,
                         int32_t C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * bf16 data path (BASELINE.json configs 4 / 5: e2fgvi_hq at 720p / 1080p, "bf16 MFMA").
 * Activations live in HBM as bf16 NHWC (element (n,y,x,c) at ((n*H+y)*W+x)*ld + c, ld in ELEMENTS), weights are packed
 * bf16, every product runs on v_mfma_f32_32x32x16_bf16 with fp32 accumulation; bias / residual / activation are applied
 * in fp32 and the result is stored as bf16 and / or fp32.  Same operators and call sites as e2fgvi_conv2d_nhwc.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const void* src[E2FGVI_MAX_SRC];   /* bf16 NHWC sources of the virtual concat (fp32 for e2fgvi_conv2d_f32x) */
    int32_t src_ld[E2FGVI_MAX_SRC];    /* pixel stride, elements (multiple of 8)                           */
    int32_t src_coff[E2FGVI_MAX_SRC];  /* first channel used by group 0 (multiple of 8)                    */
    int32_t src_cpg[E2FGVI_MAX_SRC];   /* channels per group taken from this source (multiple of 8)        */
    int32_t nsrc;
    int32_t N, H, W, Ho, Wo;
    int32_t KH, KW, stride, pad;
    int32_t groups;
    int32_t Cout;
    const void* wpacked;               /* e2fgvi_pack_conv_weight_bf16x                                     */
    const float* bias;                 /* fp32 [Cout] or NULL                                               */
    const void* residual;              /* NHWC [N,Ho,Wo,*] of res_dtype, or the fp32 [P,4] flows of ACT_DCNPOST */
    int32_t res_ld, res_coff, res_dtype;
    void* dst;                         /* NHWC [N,Ho,Wo,dst_ld] of dst_dtype                                */
    int32_t dst_ld, dst_coff, dst_dtype;
    void* dst2;                        /* optional second copy of the result as bf16 NHWC, or NULL          */
    int32_t dst2_ld, dst2_coff;
    int32_t act;
    float slope;
    int32_t tile;                      /* 0 = auto; rows x columns per workgroup: 1 = 128x128, 2 = 128x64, 3 = 128x32,
                                        * 4 = 64x128, 5 = 64x64, 6 = 256x128, 7 = 256x256, 8 = 256x192; + 10 (11-14, 16-18):
                                        * 3x3 stride-1 layers with one A stage per kernel row (bf16 operands only)      */
    int32_t dst_nchw;                  /* 1: dst is plain fp32 NCHW [N,Cout,Ho,Wo] (dst_ld / dst_coff ignored)  */
    int32_t tap_packed;                /* 1: wpacked comes from e2fgvi_pack_conv_weight_bf16x_taps (ABI version 3)  */
    /* ABI version 6 (zero = the behaviour of version 5).  out_grid = 1: Ho x Wo are taken as given, `pad` rows lie above and
     * `pad_left` columns left of the image and whatever else the KH x KW kernel reaches reads as zeros; with out_sy / out_sx > 0
     * output pixel (n, oy, ox) is stored -- and the residual read -- at pixel (oy*out_sy + out_py, ox*out_sx + out_px) of image n
     * of an [N, out_H, out_W] tensor.  res_bcast = 1: `residual` is ONE image ([out_H*out_W] or [Ho*Wo] rows) added to every
     * image of the batch.  Together: SoftComp in gather form -- nn.Fold(7x7, stride 3, pad 3) of Linear(512 -> 49*128)
     * (tfocal_transformer.py:49-72, tfocal_transformer_hq.py:49-79) as nine phase convolutions over the token grid, phase
     * (py, px) owning the pixels (3 ty + py, 3 tx + px): no [tokens, 6272] tensor (813 MB at 720p T=10 in bf16). */
    int32_t out_grid, pad_left;
    int32_t out_sy, out_sx, out_py, out_px, out_H, out_W;
    int32_t res_bcast;
    /* ABI version 8 (zero = the behaviour of version 7).  dst2_plane_stride > 0 (fp32 dst, groups = 1, no NCHW / scatter):
     * output channels co >= dst2_split_from are NOT stored to dst; their EXACT three-way bf16 split (hi + mid + lo == the fp32
     * result, csrc/common.h e2_split2) goes to dst2 instead -- plane pl (0 = hi, 1 = mid, 2 = lo) of row m, channel co at
     * dst2[pl * dst2_plane_stride + m * dst2_ld + dst2_coff + co - dst2_split_from] (elements) -- and channels below
     * dst2_split_from go to dst only.  The qkv Linear of a transformer block writes the K / V operand planes of the
     * split-operand attention (e2fgvi_focal_attention_x3) this way: no separate e2fgvi_split3_kv pass over the rows
     * (tfocal_transformer.py:221-223: q = columns 0-511, k = 512-1023, v = 1024-1535). */
    int32_t dst2_split_from;
    int64_t dst2_plane_stride;
} e2fgvi_convx_desc;

int e2fgvi_conv2d_bf16x(const e2fgvi_convx_desc* d, void* stream);
/* number of bf16 elements of the packed weight buffer: [group][K-step][8 k-octets][Npad][8], a K-step = 64 input
 * channels of one (tap, source), sources padded to 64, Npad = Cout/groups rounded up to 32 */
int64_t e2fgvi_packed_conv_weight_bf16x_size(int32_t Cout, int32_t groups, int32_t KH, int32_t KW, int32_t nsrc,
                                             const int32_t* src_cpg);
/* w: fp32 [Cout, sum(cpg), KH, KW] (torch OIHW) */
int e2fgvi_pack_conv_weight_bf16x(const float* w, void* wpacked, int32_t Cout, int32_t groups, int32_t KH, int32_t KW,
                                  int32_t nsrc, const int32_t* src_cpg, void* stream);

/* Tap-packed weights for narrow layers (ONE bf16 source of 8 ... 56 channels, no groups, KW >= 2: SPyNet's 7x7 stacks
 * model/modules/flow_comp.py:180-215, the encoder's first layer e2fgvi.py:76, the FFN's second Linear read as a 7x7 stride-3
 * convolution of the folded tensor tfocal_transformer.py:81,95-97): the (tap, 8-channel chunk) pairs form one stream cut into
 * K-steps of 8 chunks -- 49 taps of 8 / 16 / 32 / 40 channels in 7 / 13 / 25 / 31 steps instead of 49 zero-padded ones.
 * Set tap_packed = 1 in the descriptor; the row-shift tile codes do not apply.  The _f32x_ pair is the same layout for
 * e2fgvi_conv2d_f32x (chunks of 4 fp32 channels, K-steps of 32: the fp32 path's FFN second Linear, 40 channels = 10 chunks
 * per tap, 62 steps instead of 98). */
int64_t e2fgvi_packed_conv_weight_bf16x_taps_size(int32_t Cout, int32_t KH, int32_t KW, int32_t cin);
int e2fgvi_pack_conv_weight_bf16x_taps(const float* w, void* wpacked, int32_t Cout, int32_t KH, int32_t KW, int32_t cin,
                                       void* stream);
int64_t e2fgvi_packed_conv_weight_f32x_taps_size(int32_t Cout, int32_t KH, int32_t KW, int32_t cin);
int e2fgvi_pack_conv_weight_f32x_taps(const float* w, float* wpacked, int32_t Cout, int32_t KH, int32_t KW, int32_t cin,
                                      void* stream);

/* The same LDS-DMA kernel on fp32 operands: fp32 NHWC sources (channels per source in multiples of 4), fp32 packed weights,
 * v_mfma_f32_32x32x2_f32 (exact fp32, a K-step = 32 channels).  Same descriptor; used by the fp32 path for its GEMM-shaped
 * layers (token Linears, SoftSplit / SoftComp) where it beats e2fgvi_conv2d_nhwc's register-staged pipeline. */
int e2fgvi_conv2d_f32x(const e2fgvi_convx_desc* d, void* stream);
int64_t e2fgvi_packed_conv_weight_f32x_size(int32_t Cout, int32_t groups, int32_t KH, int32_t KW, int32_t nsrc,
                                            const int32_t* src_cpg);
int e2fgvi_pack_conv_weight_f32x(const float* w, float* wpacked, int32_t Cout, int32_t groups, int32_t KH, int32_t KW,
                                 int32_t nsrc, const int32_t* src_cpg, void* stream);

/* ABI version 7.  fp32 layers on the bf16 matrix pipe by exact operand splitting ("x3"): fp32 NHWC sources as for
 * e2fgvi_conv2d_f32x; every weight is stored as three bf16 numbers whose sum is the fp32 weight bit for bit (hi / mid / lo,
 * 8 significand bits each), every activation is split the same way in registers, and of the nine bf16 products of a*b the six
 * largest are accumulated by v_mfma_f32_32x32x16_bf16 in fp32 (the three dropped ones are < 2^-22 |a*b| together): fp32-level
 * rounding at 2.7x the fp32 MFMA rate.  Same descriptor, tile codes 1..8 and 107 / 108 (the 256x256 / 256x192 tiles with the two
 * halves of the workgroup one K-half-step apart: same bits as 7 / 8, not for tap-packed weights); wpacked holds 3 x the fp32 element count, in bf16
 * ([group][K-step][plane][4 k-octets][Npad][8]).  Replaces the same reference calls as e2fgvi_conv2d_nhwc (nn.Conv2d /
 * nn.Linear of model/e2fgvi_hq.py, model/modules/tfocal_transformer_hq.py). */
int e2fgvi_conv2d_f32x3(const e2fgvi_convx_desc* d, void* stream);
int64_t e2fgvi_packed_conv_weight_f32x3_size(int32_t Cout, int32_t groups, int32_t KH, int32_t KW, int32_t nsrc,
                                             const int32_t* src_cpg);
int e2fgvi_pack_conv_weight_f32x3(const float* w, void* wpacked, int32_t Cout, int32_t groups, int32_t KH, int32_t KW,
                                  int32_t nsrc, const int32_t* src_cpg, void* stream);
/* ... and the Winograd F(2x2,3x3) kernel the same way (csrc/conv_wino.hip, X3 build): the transformed input and the
 * transformed weights are split exactly into three bf16 pieces each, six bf16 MFMA terms per product.  Same descriptor and
 * epilogues as e2fgvi_conv3x3_winograd; tile 0 (auto), 32, 132, 164; wpacked holds
 * e2fgvi_packed_winograd_weight_x3_size() bf16 elements ([group][16-channel stage][16 positions][plane][h][Npad][8]). */
int e2fgvi_conv3x3_winograd_x3(const e2fgvi_conv_desc* d, void* stream);
int64_t e2fgvi_packed_winograd_weight_x3_size(int32_t Cout, int32_t groups, int32_t nsrc, const int32_t* src_cpg);
int e2fgvi_pack_winograd_weight_x3(const float* w, void* wpacked, int32_t Cout, int32_t groups, int32_t nsrc,
                                   const int32_t* src_cpg, void* stream);
int64_t e2fgvi_packed_conv_weight_f32x3_taps_size(int32_t Cout, int32_t KH, int32_t KW, int32_t cin);
int e2fgvi_pack_conv_weight_f32x3_taps(const float* w, void* wpacked, int32_t Cout, int32_t KH, int32_t KW, int32_t cin,
                                       void* stream);

/* ABI version 7: e2fgvi_focal_attention (fp32 in, fp32 softmax, fp32 out; tfocal_transformer.py:226-396) with both matrix
 * products on the bf16 matrix pipe as six exact bf16 terms of three-way split operands (csrc/attention_x3.hip).
 * e2fgvi_split3_kv: the k / v columns (512 .. 1535) of `rows` consecutive fp32 qkv rows -- the B*T*fh*fw token rows FOLLOWED by
 * the B*T*nWin pooled rows -- as three bf16 planes planes[3][rows][1024] whose sum is the fp32 value bit for bit.
 * e2fgvi_focal_attention_x3: qkv = the token rows (read for Q), planes = that buffer; waves: 0 (auto), 2, 4, 8 = waves of 32
 * queries per workgroup, 14 = four waves x two key groups (small grids: fewer than two query blocks per SIMD). */
int e2fgvi_split3_kv(const float* qkv_rows, void* planes, int64_t rows, void* stream);
int e2fgvi_focal_attention_x3(const float* qkv, const void* planes, const int32_t* key_tab, int32_t tab_ld, const int32_t* nkeys,
                              float* out, int32_t B, int32_t T, int32_t fh, int32_t fw, int32_t waves, void* stream);

/* Fused temporal focal window attention on bf16 MFMA: qkv / kv_pool / out are bf16 with the layouts of
 * e2fgvi_focal_attention; scores, softmax statistics and accumulation are fp32.  qkv and kv_pool must lie within one
 * 4 GiB window (the engine allocates them back to back). */
int e2fgvi_focal_attention_bf16(const void* qkv, const void* kv_pool, const int32_t* key_tab, int32_t tab_ld,
                                const int32_t* nkeys, void* out, int32_t B, int32_t T, int32_t fh, int32_t fw,
                                void* stream);
/* Kernel variant of e2fgvi_focal_attention_bf16 for the following calls of this process (A/B measurements and the tests of
 * every instantiation): 0 = automatic, 1 = round 2's register-staged kernel, 10 * QB + NW = the LDS-DMA kernel with NW
 * (2 / 4 / 8) waves of QB (1 / 2) x 32 queries per workgroup.  Returns the previous setting (-1: environment default,
 * E2FGVI_ATT_VARIANT).  Same operator and results up to fp32 summation order in every variant. */
int e2fgvi_focal_attention_bf16_variant(int variant);

/* Typed variants of the HBM-bound helpers for the bf16 data path: same operators and reference call sites as the fp32
 * entry points above, tensors marked `void*` are fp32 or bf16 as the dtype argument says; all arithmetic is fp32. */
int e2fgvi_nchw_to_nhwc_x(const float* src, void* dst, int32_t dst_dtype, int32_t N, int32_t C, int32_t H, int32_t W,
                          int32_t ld, float scale, float shift, void* stream);
int e2fgvi_resize_bilinear_bf16(const void* src, int32_t src_ld, void* dst, int32_t dst_ld, int32_t N, int32_t C, int32_t H,
                                int32_t W, int32_t Ho, int32_t Wo, int32_t align_corners, void* stream);
/* flows8_bf16 (optional): the [P,4] flows again as a bf16 [P,8] conv source (channels 4..7 zero) */
int e2fgvi_prop_cond_x(const float* feat_prop, int32_t fp_ld, const float* feat_n2, int32_t f2_ld, const float* flow_a,
                       const float* flow_b, int64_t flow_img_stride, void* cond, int32_t cond_dtype, float* flows,
                       void* flows8_bf16, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* ... with the warp sources of src_dtype: E2FGVI_F32 (= e2fgvi_prop_cond_x) or E2FGVI_BF16 (bf16 NHWC features, bf16 cond:
 * half the gather bytes; the bf16 path warps the bf16 copies of the propagated features) */
int e2fgvi_prop_cond_xs(const void* feat_prop, int32_t fp_ld, const void* feat_n2, int32_t f2_ld, int32_t src_dtype,
                        const float* flow_a, const float* flow_b, int64_t flow_img_stride, void* cond, int32_t cond_dtype,
                        float* flows, void* flows8_bf16, int32_t N, int32_t H, int32_t W, int32_t C, void* stream);
/* out_bf16 (optional): the 8 input channels again as bf16 [Np,h,w,8], the source of the level's bf16 conv stack */
int e2fgvi_spynet_level_input_x(const float* pyr, const int32_t* ref_idx, const int32_t* supp_idx, const float* flow_prev,
                                float* out, void* out_bf16, int32_t Np, int32_t h, int32_t w, void* stream);
int e2fgvi_layernorm_x(const float* x, const float* gamma, const float* beta, void* y, int32_t y_dtype, int64_t rows,
                       int32_t C, void* stream);
int e2fgvi_window_pool_x(const void* x, int32_t dtype, const float* w45, const float* bias1, void* pooled, int32_t BT,
                         int32_t fh, int32_t fw, int32_t C, void* stream);
int e2fgvi_ffn_fold_x(const void* hid, void* folded, int32_t dtype, int32_t F, int32_t fh, int32_t fw, int32_t H, int32_t W,
                      int32_t C, void* stream);
int e2fgvi_ffn_unfold_gelu_x(const void* folded, void* out, int32_t dtype, int32_t F, int32_t fh, int32_t fw, int32_t H,
                             int32_t W, int32_t C, void* stream);
/* The FFN middle with the GELU in front of the unfold (a gather with zero padding commutes with GELU, GELU(0) = 0):
 * folded = GELU(fold(hid) / count) [F,H,W,C], then a pure unfold -- 5.4x fewer erf evaluations than ffn_unfold_gelu
 * (tfocal_transformer.py:82,92-97).  fp32: bit-identical to the pair above; bf16: GELU sees the unrounded fold. */
int e2fgvi_ffn_fold_gelu_x(const void* hid, void* folded, int32_t dtype, int32_t F, int32_t fh, int32_t fw, int32_t H, int32_t W,
                           int32_t C, void* stream);
int e2fgvi_ffn_unfold_x(const void* folded, void* out, int32_t dtype, int32_t F, int32_t fh, int32_t fw, int32_t H, int32_t W,
                        int32_t C, void* stream);
int e2fgvi_softcomp_fold_bf16(const void* emb, const float* bias_hwc, const void* residual, void* dst, int32_t F, int32_t fh,
                              int32_t fw, int32_t H, int32_t W, int32_t C, void* stream);
/* element-wise fp32 <-> bf16 conversion (round to nearest even), n a multiple of 4 */
int e2fgvi_cast(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* E2FGVI_HIP_H */
