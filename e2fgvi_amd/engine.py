"""Device-side execution plan of the E2FGVI / E2FGVI-HQ inference forward on MI355X.

``Engine`` is built once from a checkpoint-format ``state_dict`` (reference format):
it re-lays-out every weight for the HIP kernels and then runs the forward as a sequence of
C-ABI kernel launches (ops.py).  The staging mirrors the reference's module boundaries so each
stage can be checked against the oracle on the same inputs:

    flows()      e2fgvi.py:210-234 + flow_comp.py:84-169      SPyNet, both directions batched
    encode()     e2fgvi.py:96-109                             9 convs, grouped concat by pointers
    propagate()  feat_prop.py:81-149, :35-58                  2 x l_t sequential steps
    transformer()tfocal_transformer.py:466-536, :210-399      8 blocks, fused focal attention
    compose()    tfocal_transformer.py:65-72 (+ e2fgvi.py:258)
    decode()     e2fgvi.py:126-150, :261-262

All activations are NHWC fp32; nothing here computes on the CPU or through torch ops -- torch only
owns the buffers (and does index-only copies when clips are batched).
"""
import numpy as np
import os

import torch

from . import ops
from .engine_x import BF16Path
from .ops import ACT_DCNPOST, ACT_LRELU, ACT_NONE, ACT_RELU, ACT_TANH, PackedConv, PackedConvX, PackedDcn, PackedLinear

# fp32 path: the FFN's second Linear as a conv of the folded tensor (as the bf16 path runs it).  Measured neutral on the fp32
# MFMA kernels (15.785 vs 15.775 ms, profiles/r02_fc2_conv.txt: the 16-byte tap-packed fetches cost what the unfold kernel
# saved); with the split-operand kernels (ops.X3_ENABLED) the conv form wins -- 743 -> 750 frames/s, same box, two runs each
# (profiles/r03_fc2_conv_x3.txt) -- and is the default
FC2_CONV = ops.X3_ENABLED          # (bench.py's E2FGVI_X3=0 secondary line sets both to False for its engine)
WIN = (5, 9)
# where the side stream (SPyNet) joins the main one: in front of encoder.layers.<JOIN_AT> (18 = behind the encoder).  Round 4 joined in
# front of layer 10 (its wide-tile kernel was fenced to layers with the chip to themselves); with encoder.layers.2 / 6 / 8 on that
# kernel the main stream reaches layer 10 before SPyNet has finished and waited there: 787.9 frames/s joined at 10, 801-808 joined
# at 12 ... 18, best at 16 (same box, two runs each: profiles/r05_join_position.txt)
# With several clips per forward the batched encoder layers fill the chip on their own and the early join is 0.5 % ahead (8 clips: 974.7
# vs 969.5 frames/s), so: 16 at one clip, 10 otherwise; E2FGVI_JOIN_AT overrides both.
JOIN_AT = int(os.environ.get("E2FGVI_JOIN_AT", "0") or 0)
if JOIN_AT not in (0, 10, 12, 14, 16, 18):
    raise ValueError("E2FGVI_JOIN_AT=%d: the join sits in front of encoder.layers.10 / 12 / 14 / 16 or behind the encoder (18)" % JOIN_AT)
PROP_SPLIT = os.environ.get("E2FGVI_PROP_SPLIT", "1") != "0"       # 0: conv_offset.0 / backbone.0 whole in every propagation step (A/B)


def token_grid(h, w):
    """Unfold(7, stride 3, pad 3) output grid (tfocal_transformer.py:30-37)."""
    return (h + 6 - 7) // 3 + 1, (w + 6 - 7) // 3 + 1


def build_key_table(fh, fw, valid_ind_rolled):
    """Per-window key references for the fused attention kernel.

    Row ``win`` lists, for one frame, the keys window ``win`` attends to besides nothing else:
    its own 45 tokens, the 120 ring tokens of the four circularly rolled maps (positions from
    ``valid_ind_rolled``; wrap-around and the 12 duplicates kept exactly as torch.roll +
    window_partition produce them, tfocal_transformer.py:235-273) as ``y*fw+x`` >= 0, then the
    valid pooled windows of the 5x9 neighbourhood (:328-333) as ``-(index+1)``.  ``nkeys`` is the
    count; the other pooled slots are the zero-padded ones (score -100).
    """
    nwh, nww = fh // WIN[0], fw // WIN[1]
    ex = (WIN[0] // 2, WIN[1] // 2)
    shifts = ((-ex[0], -ex[1]), (-ex[0], ex[1]), (ex[0], -ex[1]), (ex[0], ex[1]))   # tl, tr, bl, br
    vi = [int(v) for v in valid_ind_rolled]
    tab = np.zeros((nwh * nww, 212), np.int32)
    nk = np.zeros((nwh * nww,), np.int32)
    for wy in range(nwh):
        for wx in range(nww):
            refs = []
            for p in range(45):
                refs.append((wy * 5 + p // 9) * fw + wx * 9 + p % 9)
            for v in vi:
                n, p = divmod(v, 45)
                sy, sx = shifts[n]
                # rolled[y,x] = k[(y - sy) % fh, (x - sx) % fw]
                y = (wy * 5 + p // 9 - sy) % fh
                x = (wx * 9 + p % 9 - sx) % fw
                refs.append(y * fw + x)
            for ki in range(5):
                for kj in range(9):
                    py, px = wy - 2 + ki, wx - 4 + kj
                    if 0 <= py < nwh and 0 <= px < nww:
                        refs.append(-(py * nww + px + 1))
            w = wy * nww + wx
            nk[w] = len(refs)
            tab[w, :len(refs)] = refs
    return tab, nk


def split_prop_weights(w_off0, w_bb0, ch=128):
    """The propagation split as weight slices.  conv_offset.0 reads cat(cond_n1, current frame, cond_n2, flow_1,
    flow_2) (feat_prop.py:27-35,117-123) and backbone.0 cat(current frame[, the other direction's feature], feat_prop)
    (feat_prop.py:131-137): `*_rec` keep the input channels that depend on the recurrence -- sources (cond_n1, cond_n2, flows) and
    (feat_prop) --, `off_cur` / `bb_pre` the rest; conv(whole) == conv(rec part) + conv(other part), bias counted once
    (tests/test_identities.py)."""
    return dict(off_rec=torch.cat([w_off0[:, :ch], w_off0[:, 2 * ch:]], 1).contiguous(), off_cur=w_off0[:, ch:2 * ch].contiguous(),
                bb_rec=w_bb0[:, -ch:].contiguous(), bb_pre=w_bb0[:, :-ch].contiguous())


class _NotBuilt:
    """stands in for an fp32 layer in the bf16 mode (names / flags may be set on it; calling it is a bug)"""

    def __init__(self, *a, **k):
        self.name = "not built"

    def __call__(self, *a, **k):
        raise RuntimeError("fp32 layer %r was not built: this engine runs the bf16 data path" % self.name)


class Engine(BF16Path):
    def __init__(self, state_dict, model="e2fgvi", device="cuda", precision="fp32", winograd=True, autotune=True):
        """precision="fp32": fp32 tensors, every contraction with fp32-level rounding (the default and the parity configuration):
        on the fp32 MFMA instructions (bit-equivalent to an fp32 FMA chain) or -- ops.X3_ENABLED, the default -- on the bf16 matrix
        pipe with exactly split operands (three bf16 pieces per fp32 value, six of the nine cross terms, fp32 accumulation: fp32-
        level, not bit-identical to an FMA chain; tests/test_gpu_x3.py).  Which of the two a layer runs comes from the decision
        table (ops.py); E2FGVI_X3=0 gives the pure fp32-MFMA configuration.
        precision="bf16": the bf16 data path of engine_x.py (BASELINE.json HQ configurations): bf16 activations in HBM,
        every conv / linear / attention product on bf16 MFMA with fp32 accumulation; SPyNet, the flows, the DCN offsets
        and masks, the deformable conv's arithmetic and the token residual stream stay fp32."""
        if precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16'")
        self.precision = precision
        self.hq = model == "e2fgvi_hq"
        self.device = torch.device(device)
        sd = {k: v.detach().to(self.device) for k, v in state_dict.items()}
        self.sd = sd
        f = lambda k: sd[k].float().contiguous()
        self.bf16 = precision == "bf16"
        # bf16 mode: the fp32 layers are never called (forward_x reads only the LayerNorm / pooling parameters, the key
        # tables and sc.bias from this constructor) -- do not pack their weights: several hundred MB of device memory (both
        # the implicit-GEMM and the Winograd packings) and the start-up time of ~150 pack launches
        PackedConv, PackedLinear, PackedDcn, PackedConvX, PackedTail = (
            (_NotBuilt,) * 5 if self.bf16 else (ops.PackedConv, ops.PackedLinear, ops.PackedDcn, ops.PackedConvX, ops.PackedTailConv))
        pw = dict(precision="fp32")
        # wide 3x3 / stride-1 layers: fp32 Winograd F(2x2,3x3) whenever the call qualifies (even H, W), else implicit GEMM
        ww = dict(precision="fp32", algo="auto" if winograd else "igemm")

        # ---- encoder (e2fgvi.py:75-94)
        w0 = torch.zeros(64, 4, 3, 3, device=self.device)
        w0[:, :3] = f("encoder.layers.0.weight")
        enc = [PackedConv(w0, f("encoder.layers.0.bias"), [4], stride=2, pad=1)]
        for i, (cpg, g, s) in zip((2, 4, 6, 8, 10, 12, 14, 16),
                                  (([64], 1, 1), ([64], 1, 2), ([128], 1, 1), ([256], 1, 1), ([128, 192], 2, 1),
                                   ([64, 128], 4, 1), ([32, 48], 8, 1), ([256, 256], 1, 1))):
            enc.append(PackedConv(f("encoder.layers.%d.weight" % i), f("encoder.layers.%d.bias" % i), cpg, groups=g,
                                  stride=s, pad=1, **(ww if s == 1 else pw)))
        self.enc = enc

        # ---- decoder (e2fgvi.py:143-150)
        self.dec = [PackedConv(f("decoder.0.conv.weight"), f("decoder.0.conv.bias"), [128], pad=1, **ww),
                    PackedConv(f("decoder.2.weight"), f("decoder.2.bias"), [128], pad=1, **ww),
                    PackedConv(f("decoder.4.conv.weight"), f("decoder.4.conv.bias"), [64], pad=1, **ww),
                    PackedTail(f("decoder.6.weight"), f("decoder.6.bias"))]

        # ---- propagation (feat_prop.py:61-79, :15-33)
        # conv_offset.0 and backbone.0 are linear in their input channels, and a third / a half / two thirds of those do not
        # depend on the recurrence (the current frame's features, the other direction's result): at one clip per forward that
        # part is computed for all frames at once on the side stream, beside the chain of one-frame launches that leaves most
        # of the chip idle, and enters the per-step layer -- now 260 / 128 input channels instead of 388 / 256 / 384 -- as its
        # residual (PROP_SPLIT, propagate()).  Round 2 measured the same split as a net loss (42 -> 34 and 37 -> 26 us per step
        # against 537 us of batched launches in the critical path, profiles/r02_c2_layer_fp32_base.md): the batched half was
        # four fp32-MFMA launches on the main stream then, it is three split-operand launches beside the chain now.
        self.prop = {}
        self.prop_split = {}
        for d, nparts in (("backward_", 2), ("forward_", 3)):
            p = "feat_prop_module.deform_align.%s." % d
            # conv_offset.0 input = cat(cond_n1, cur, cond_n2, flow_1, flow_2): sources (cond|0), cur, (cond|128), flows4
            off = [PackedConv(f(p + "conv_offset.0.weight"), f(p + "conv_offset.0.bias"), [128, 128, 128, 4], pad=1, **ww),
                   PackedConv(f(p + "conv_offset.2.weight"), f(p + "conv_offset.2.bias"), [128], pad=1, **ww),
                   PackedConv(f(p + "conv_offset.4.weight"), f(p + "conv_offset.4.bias"), [128], pad=1, **ww),
                   PackedConv(f(p + "conv_offset.6.weight"), f(p + "conv_offset.6.bias"), [128], pad=1,
                              algo=ww["algo"])]
            # (split-operand MFMA with the other x3 kernels: ops.X3_ENABLED)
            dcn = PackedDcn(f(p + "weight"), f(p + "bias"), 16, pad=1, mfma="x3" if (precision == "fp32" and ops.X3_ENABLED) else "fp32")
            b = "feat_prop_module.backbone.%s." % d
            bb = [PackedConv(f(b + "0.weight"), f(b + "0.bias"), [128] * nparts, pad=1, **ww),
                  PackedConv(f(b + "2.weight"), f(b + "2.bias"), [128], pad=1, **ww)]
            self.prop[d] = (off, dcn, bb)
            if PROP_SPLIT and precision == "fp32":
                ws = split_prop_weights(f(p + "conv_offset.0.weight"), f(b + "0.weight"))
                sp = dict(off_rec=PackedConv(ws["off_rec"], f(p + "conv_offset.0.bias"), [128, 128, 4], pad=1, **ww),
                          off_cur=PackedConv(ws["off_cur"], None, [128], pad=1, **ww),
                          bb_rec=PackedConv(ws["bb_rec"], f(b + "0.bias"), [128], pad=1, **ww),
                          bb_pre=PackedConv(ws["bb_pre"], None, [128] * (nparts - 1), pad=1, **ww))
                sp["off_rec"].name, sp["off_cur"].name = "deform_align.%sconv_offset.0 (recurrent part)" % d, "deform_align.%sconv_offset.0 (current-frame part)" % d
                sp["bb_rec"].name, sp["bb_pre"].name = "backbone.%s0 (recurrent part)" % d, "backbone.%s0 (non-recurrent part)" % d
                self.prop_split[d] = sp
        self.fusion = PackedConv(f("feat_prop_module.fusion.weight"), f("feat_prop_module.fusion.bias"), [128, 128], **pw)

        # ---- soft split / composite (tfocal_transformer.py:19-72)
        self.ss = PackedConv(f("ss.embedding.weight").view(512, 128, 7, 7), f("ss.embedding.bias"), [128], stride=3, pad=3,
                             **pw)
        # patch channel order c*49+tap -> tap*128+c (private layout of the fold kernels)
        wsc = f("sc.embedding.weight").view(128, 49, 512).permute(1, 0, 2).reshape(6272, 512).contiguous()
        bsc = f("sc.embedding.bias").view(128, 49).t().reshape(6272).contiguous()
        self.sc = PackedLinear(wsc, bsc, **pw)
        if self.hq:
            self.sc_bias_conv = PackedConv(f("sc.bias_conv.weight"), f("sc.bias_conv.bias"), [128], pad=1, **ww)
            self.sc_bias_hwc = None
        else:
            self.sc_bias_hwc = f("sc.bias").permute(1, 2, 0).contiguous()

        # ---- transformer blocks
        self.blocks = []
        for i in range(8):
            p = "transformer.%d." % i
            w1 = f(p + "mlp.conv1.0.weight").view(40, 49, 512).permute(1, 0, 2).reshape(1960, 512).contiguous()
            b1 = f(p + "mlp.conv1.0.bias").view(40, 49).t().reshape(1960).contiguous()
            w2 = f(p + "mlp.conv2.1.weight").view(512, 40, 49).permute(0, 2, 1).reshape(512, 1960).contiguous()
            self.blocks.append(dict(
                pool_w=f(p + "pool_layers.0.weight").view(45), pool_b=f(p + "pool_layers.0.bias"),
                n1w=f(p + "norm1.weight"), n1b=f(p + "norm1.bias"), n2w=f(p + "norm2.weight"), n2b=f(p + "norm2.bias"),
                qkv=PackedLinear(f(p + "attn.qkv.weight"), f(p + "attn.qkv.bias"), **pw),
                proj=PackedLinear(f(p + "attn.proj.weight"), f(p + "attn.proj.bias"), **pw),
                fc1=PackedLinear(w1, b1, **pw),
                # fc2: Linear(1960 -> 512) of the unfolded 7x7 patches == the 7x7 / stride 3 / pad 3 convolution of the folded
                # [F, H, W, 40] tensor (tfocal_transformer.py:81,95-97): no unfold kernel, no [rows, 1960] tensor
                fc2=(PackedConvX(f(p + "mlp.conv2.1.weight").view(512, 40, 7, 7), f(p + "mlp.conv2.1.bias"), [40], stride=3,
                                 pad=3, dtype=torch.float32, taps=True) if FC2_CONV and precision == "fp32" else
                     PackedLinear(w2, f(p + "mlp.conv2.1.bias"), **pw)),
                valid=sd[p + "attn.valid_ind_rolled"].cpu().tolist()))

        # ---- SPyNet (flow_comp.py:49-82,172-215)
        self.spy = []
        for lv in range(6):
            convs = []
            for j, cin in enumerate((8, 32, 64, 32, 16)):
                p = "update_spynet.basic_module.%d.basic_module.%d.conv." % (lv, j)
                convs.append(PackedConv(f(p + "weight"), f(p + "bias"), [cin], pad=3))
                convs[-1].nopk = True      # SPyNet runs on a side stream: the build without packed-fp32 VALU (build.py)
            self.spy.append(convs)
        mean = f("update_spynet.mean").view(3)
        std = f("update_spynet.std").view(3)
        one = torch.ones(1, device=self.device)
        self.spy_scale = torch.cat([1.0 / std, one]).contiguous()
        self.spy_shift = torch.cat([-mean / std, 0 * one]).contiguous()
        self.half = torch.full((4,), 0.5, device=self.device)
        self._tables = {}
        self._zeros = {}
        # checkpoint names on the layers (launch traces: tools/layer_table.py, bench.py's FLOP accounting)
        for k, i in enumerate((0, 2, 4, 6, 8, 10, 12, 14, 16)):
            self.enc[k].name = "encoder.layers.%d" % i
        for k, n in enumerate(("decoder.0.conv", "decoder.2", "decoder.4.conv", "decoder.6")):
            self.dec[k].name = n
        for d, (off, dcn, bb) in self.prop.items():
            for k, c in enumerate(off):
                c.name = "deform_align.%sconv_offset.%d" % (d, 2 * k)
            dcn.name = "deform_align.%sdcn" % d
            bb[0].name, bb[1].name = "backbone.%s0" % d, "backbone.%s2" % d
        self.fusion.name, self.ss.name, self.sc.name = "fusion", "ss.embedding", "sc.embedding"
        if self.hq:
            self.sc_bias_conv.name = "sc.bias_conv"
        for i, blk in enumerate(self.blocks):
            for k in ("qkv", "proj", "fc1", "fc2"):
                blk[k].name = "transformer.%d.%s" % (i, k)
        for lv, convs in enumerate(self.spy):
            for j, c in enumerate(convs):
                c.name = "spynet.%d.%d" % (lv, j)
        if autotune and precision == "fp32":
            # GEMM-shaped layers (token Linears, soft split / composite): the best implicit-GEMM tile depends on the
            # token count; their tile code comes from the decision table of ops.py (e2fgvi_amd/tile_table.py by default,
            # timed on the first eager call of each size class under E2FGVI_AUTOTUNE=1)
            for blk in self.blocks:
                for k in ("qkv", "proj", "fc1", "fc2"):
                    blk[k].tune = True
            self.ss.tune = self.sc.tune = self.fusion.tune = True
            # (Winograd block shapes are NOT tuned at run time: measured in round 2, the timing-based choice between the
            # 16x16 / 8x16-pixel blocks moved the forward by -1 ... -2 % and added run-to-run variance; the static rule of
            # e2fgvi_conv3x3_winograd stays.)
        if precision == "fp32" and ops.X3_ENABLED:
            # Every fp32 conv / linear of the MAIN stream may run on the bf16 matrix pipe instead (ops.PackedConvX x3: operands
            # split exactly into three bf16 pieces, six bf16 MFMA terms per product, fp32-level rounding): timed against the
            # layer's fp32 kernel on the first eager call of each size class, kept where it is faster (the GEMM-shaped layers:
            # token Linears, soft split / composite, the stride-2 and 1x1 convs; the Winograd layers mostly keep Winograd).
            # SPyNet's 7x7 layers are candidates too since round 5 (the split-operand GEMM is built without packed-fp32 VALU like every
            # kernel that can run on the side stream): the two wide layers of the upper levels take it, 221 -> 199 and 239 -> 223 us
            # alone and -- what matters beside the encoder -- at a third of the matrix-pipe time (profiles/r05_spynet_x3.txt).
            for layer in self.enc + self.dec[:3] + [self.fusion, self.ss, self.sc] + ([self.sc_bias_conv] if self.hq else []):
                layer.try_x3 = True
            for convs in self.spy:
                for layer in convs:
                    layer.try_x3 = True
            for off, _dcn, bb in self.prop.values():
                for layer in off + bb:
                    layer.try_x3 = True
            for sp in self.prop_split.values():
                for layer in sp.values():
                    layer.try_x3 = True
            for blk in self.blocks:
                for k in ("qkv", "proj", "fc1", "fc2"):
                    blk[k].try_x3 = True
        # SPyNet runs on a side stream next to the encoder, in both precision modes.  Round 1 found the side stream's
        # kernels corrupted beside bf16 MFMA tiles; round 2 traced it to packed-fp32 VALU instructions consuming freshly
        # loaded registers (tools/probe/overlap_probe.hip, DESIGN.md "Stream overlap"): every kernel that can run on the
        # side stream is built without them (csrc/misc.hip and the `nopk` build of conv.hip, e2fgvi_amd/build.py).
        self.overlap_flows = True
        if self.bf16:
            self.autotune_x = autotune
            self._init_x(f)
        self._side = None
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------ helpers
    def weight_bytes(self):
        """device bytes of the packed weights the layers hold right now (each layer packs what the kernels it has run need:
        ops.PackedConv); the checkpoint's own tensors are not counted"""
        seen, total = set(), 0

        def walk(o):
            nonlocal total
            if id(o) in seen:
                return
            seen.add(id(o))
            if isinstance(o, (list, tuple)):
                for v in o:
                    walk(v)
            elif isinstance(o, dict):
                for v in o.values():
                    walk(v)
            elif hasattr(o, "weight_bytes") and o is not self:
                total += o.weight_bytes()
            elif isinstance(getattr(o, "wpacked", None), torch.Tensor):
                total += o.wpacked.numel() * o.wpacked.element_size()
            elif hasattr(o, "__dict__") and type(o).__module__.startswith("e2fgvi_amd") and o is not self:
                walk(list(vars(o).values()))
        walk(list(vars(self).values()))
        return total

    def _side_stream(self):
        """The side stream that belongs to the CURRENT stream: one per main stream a forward has been issued on.  With a single
        shared side stream, two eager forwards on two main streams (video.inpaint_video(in_flight=2)) raced through the caching
        allocator: a tensor allocated on the side stream is returned to that stream's pool when its forward ends on the HOST, the
        other forward's side work takes the block while the first forward's main-stream kernels -- which no wait of the second
        forward covers -- are still reading it (round 6: different bytes from the one-at-a-time run; graph replay, with its private
        pools and static buffers, was never exposed)."""
        if self._side is None:
            self._side = {}
        key = torch.cuda.current_stream(self.device).cuda_stream
        if key not in self._side:
            # (default priority on purpose: with a priority on EITHER branch of the captured forward -- this stream or the capture
            #  stream -- the replayed graph takes 21.9 ms instead of 12.0, profiles/r05_tile8_priority_ab.txt)
            self._side[key] = torch.cuda.Stream(device=self.device)
        return self._side[key]

    def _zero(self, shape):
        key = tuple(shape)
        if key not in self._zeros:
            self._zeros[key] = torch.zeros(key, dtype=torch.float32, device=self.device)
        return self._zeros[key]

    def _table(self, fh, fw, blk):
        key = (fh, fw, tuple(blk["valid"]))
        if key not in self._tables:
            tab, nk = build_key_table(fh, fw, blk["valid"])
            self._tables[key] = (torch.from_numpy(tab).to(self.device), torch.from_numpy(nk).to(self.device))
        return self._tables[key]

    # ------------------------------------------------------------------ flows
    def flows(self, frames, l_t):
        """frames: [b,t,3,H,W] in [-1,1].  Returns (fwd, bwd) NHWC [b, l_t-1, h, w, 2]."""
        b, t, c, H, W = frames.shape
        h, w = H // 4, W // 4
        local = frames[:, :l_t].reshape(b * l_t, c, H, W)
        if not local.is_contiguous():
            local = local.contiguous()
        small = ops.resize_bilinear(local, (h, w), True, src_nchw=True, out_ld=4, scale=self.half, shift=self.half)
        w_up = w if w % 32 == 0 else 32 * (w // 32 + 1)
        h_up = h if h % 32 == 0 else 32 * (h // 32 + 1)
        pyr = [ops.resize_bilinear(small, (h_up, w_up), False, channels=3, out_ld=4, scale=self.spy_scale,
                                   shift=self.spy_shift)]
        for _ in range(5):
            pyr.append(ops.avgpool2(pyr[-1]))
        pyr = pyr[::-1]
        nf = b * (l_t - 1)
        key = ("pairs", b, l_t, h, w)
        if key not in self._tables:          # host->device uploads happen once (HIP-graph capture safe)
            ref, supp = [], []
            for bi in range(b):
                for i in range(l_t - 1):
                    ref.append(bi * l_t + i); supp.append(bi * l_t + i + 1)
            ref, supp = ref + supp, supp + ref
            self._tables[key] = (torch.tensor(ref, dtype=torch.int32, device=self.device),
                                 torch.tensor(supp, dtype=torch.int32, device=self.device),
                                 torch.tensor([float(w) / float(w_up), float(h) / float(h_up)], dtype=torch.float32,
                                              device=self.device))
        ref_idx, supp_idx, sc = self._tables[key]
        flow = None
        for lv in range(6):
            if self.bf16:
                flow = self.spynet_level_x(lv, pyr[lv], ref_idx, supp_idx, flow)
                continue
            inp = ops.spynet_level_input(pyr[lv], ref_idx, supp_idx, flow)
            cv = self.spy[lv]
            x = cv[0]([inp], act=ACT_RELU)
            x = cv[1]([x], act=ACT_RELU)
            x = cv[2]([x], act=ACT_RELU)
            x = cv[3]([x], act=ACT_RELU)
            flow = cv[4]([x], residual=inp, res_coff=6)
        flow = ops.resize_bilinear(flow, (h, w), False, scale=sc)
        fwd = flow[:nf].view(b, l_t - 1, h, w, 2)
        bwd = flow[nf:].view(b, l_t - 1, h, w, 2)
        return fwd, bwd

    # ------------------------------------------------------------------ encoder
    def encode(self, frames, join=None):
        """join: called in front of encoder.layers.<join_at> (16 at one clip: the fork's other branch, SPyNet, overlaps layers 0 .. 14)"""
        b, t, c, H, W = frames.shape
        x = ops.nchw_to_nhwc(frames.reshape(b * t, c, H, W).contiguous(), ld=4)
        e = self.enc
        lr = dict(act=ACT_LRELU, slope=0.2)
        x = e[0]([x], **lr)
        x = e[1]([x], **lr)
        x = e[2]([x], **lr)
        x0 = e[3]([x], **lr)
        x = e[4]([x0], **lr)
        join_at = JOIN_AT or (16 if b == 1 else 10)
        joined = join is None
        if not joined and join_at <= 10:
            join()
            joined = True
        for k in (5, 6, 7, 8):
            x = e[k]([x0, x], **lr)
            if not joined and (join_at <= 2 * k + 2 or k == 8):        # never leave the encoder with the fork open
                join()
                joined = True
        return x                                            # [b*t, h, w, 128]

    # ------------------------------------------------------------------ propagation
    def propagate(self, loc, flows_a, flows_b, inplace=False):
        """loc: [l_t, b, h, w, 128] frame-major local features.  flows_a / flows_b: NHWC [b,l_t-1,h,w,2]; they are
        bound positionally like the reference (e2fgvi.py:249-250): flows_a drives 'backward_', flows_b 'forward_'.
        Returns the propagated features [l_t, b, h, w, 128]; inplace: written over `loc` (the fusion layer's residual is `loc`
        itself: its epilogue reads a residual element and writes the same element in the same thread)."""
        l_t, b, h, w, ch = loc.shape
        dev = loc.device
        feats = {}
        zero = self._zero((b, h, w, ch))
        lk = dict(act=ACT_LRELU, slope=0.1)
        # the non-recurrent parts of conv_offset.0 / backbone.0 for the frames of steps 1 .. l_t - 1, all at once (see __init__);
        # step 0 of a direction (backbone only, needed at once) keeps the whole layer
        split = bool(self.prop_split) and b == 1 and l_t >= 3
        main = torch.cuda.current_stream()
        side = self._side_stream() if (split and self.overlap_flows) else None
        pre, ready = {}, {}
        if split:
            n1 = (l_t - 1) * b
            for k in ("off backward_", "bb backward_", "off forward_", "bb forward_"):
                pre[k] = torch.empty((l_t - 1, b, h, w, ch), dtype=torch.float32, device=dev)
            flat = lambda t: t.reshape(n1, h, w, ch)
            lb, lf = flat(loc[:l_t - 1]), flat(loc[1:])       # backward: steps 1.. are frames l_t-2 .. 0; forward: frames 1 .. l_t-1

            def beside(jobs):
                """jobs: (key, layer, sources) in the order the chain needs them; on the side stream when there is one"""
                if side is not None:
                    side.wait_stream(main)
                for key, layer, srcs in jobs:
                    if side is None:
                        layer(srcs, out=flat(pre[key]))
                        continue
                    with torch.cuda.stream(side):
                        layer(srcs, out=flat(pre[key]))
                        ready[key] = torch.cuda.Event()
                        ready[key].record(side)
            sb, sf = self.prop_split["backward_"], self.prop_split["forward_"]
            beside([("off backward_", sb["off_cur"], [lb]), ("bb backward_", sb["bb_pre"], [lb]), ("off forward_", sf["off_cur"], [lf])])

        def partial(key, slot):
            ev = ready.pop(key, None)
            if ev is not None:
                main.wait_event(ev)
            return pre[key][slot]
        for name, flows in (("backward_", flows_a), ("forward_", flows_b)):
            off_convs, dcn, bb = self.prop[name]
            sp = self.prop_split.get(name) if split else None
            if sp is not None and name == "forward_":
                beside([("bb forward_", sp["bb_pre"], [lf, flat(feats["backward_"][1:])])])
            store = torch.empty((l_t, b, h, w, ch), dtype=torch.float32, device=dev)
            order = list(range(l_t))
            if name == "backward_":
                order = order[::-1]
            img_stride = (l_t - 1) * h * w * 2
            hist = []                       # feature tensors in processing order
            feat_prop = zero
            for i, idx in enumerate(order):
                cur = loc[idx]
                if i > 0:
                    flow_a = flows[0, i - 1]
                    flow_b = flows[0, i - 2] if i > 1 else None
                    feat_n2 = hist[-2] if i > 1 else None
                    cond, fl = ops.prop_cond(feat_prop, feat_n2, flow_a, flow_b, img_stride)
                    if sp is not None:
                        slot = idx if name == "backward_" else idx - 1
                        x = sp["off_rec"]([(cond, 0), (cond, ch), fl], residual=partial("off " + name, slot), **lk)
                    else:
                        x = off_convs[0]([(cond, 0), cur, (cond, ch), fl], **lk)
                    x = off_convs[1]([x], **lk)
                    x = off_convs[2]([x], **lk)
                    # 10*tanh + flow.flip / sigmoid (feat_prop.py:38-53) applied in the epilogue of the last conv_offset
                    # layer: the deformable conv then reads finished offsets and masks
                    offs = off_convs[3]([x], residual=fl, act=ACT_DCNPOST, slope=10.0)
                    feat_prop = dcn([feat_prop, feat_n2 if feat_n2 is not None else zero], offs)
                if sp is not None and i > 0:
                    y = sp["bb_rec"]([feat_prop], residual=partial("bb " + name, idx if name == "backward_" else idx - 1), **lk)
                else:
                    srcs = [cur, feats["backward_"][idx], feat_prop] if name == "forward_" else [cur, feat_prop]
                    y = bb[0](srcs, **lk)
                feat_prop = bb[1]([y], residual=feat_prop, out=store[idx])
                hist.append(feat_prop)
            feats[name] = store
        out = self.fusion([feats["backward_"].view(l_t * b, h, w, ch), feats["forward_"].view(l_t * b, h, w, ch)],
                          residual=loc.view(l_t * b, h, w, ch), out=loc.view(l_t * b, h, w, ch) if inplace else None)
        return out.view(l_t, b, h, w, ch)

    # ------------------------------------------------------------------ transformer
    def soft_split(self, feat):
        return self.ss([feat])                               # [b*t, fh, fw, 512]

    def block(self, i, x, b, t, fh, fw, hw):
        """x: [b*t*fh*fw, 512] tokens in (b,t,y,x) order."""
        blk = self.blocks[i]
        H, W = hw
        tab, nk = self._table(fh, fw, blk)
        # LayerNorm output and the pooled windows share one buffer, and so do their qkv rows: ONE qkv GEMM for both
        # (and one buffer resource in the attention kernel)
        rows = x.shape[0]
        prow = b * t * (fh // 5) * (fw // 9)
        nbuf = torch.empty((rows + prow, 512), dtype=torch.float32, device=x.device)
        n1 = ops.layernorm(x, blk["n1w"], blk["n1b"], out=nbuf[:rows])
        ops.window_pool(n1, blk["pool_w"], blk["pool_b"], b * t, fh, fw, out=nbuf[rows:])
        if ops.attention_x3_applies(b, t, fh, fw):
            # both products on the bf16 matrix pipe (exactly split operands).  The k / v columns of all rows as three bf16 planes:
            # written by the qkv GEMM's epilogue when the split-operand GEMM runs it (round 5: no separate pass over the rows, and
            # the fp32 K / V columns are never stored), by e2fgvi_split3_kv otherwise (ops.PackedConv.__call__, kv_planes)
            planes = torch.empty((3, rows + prow, 1024), dtype=torch.bfloat16, device=x.device)
            both = blk["qkv"](nbuf, kv_planes=planes)
            att = ops.focal_attention_x3(both[:rows], planes, tab, nk, b, t, fh, fw)
        else:
            both = blk["qkv"](nbuf)
            att = ops.focal_attention(both[:rows], both[rows:], tab, nk, b, t, fh, fw)
        x1 = blk["proj"](att, residual=x)
        n2 = ops.layernorm(x1, blk["n2w"], blk["n2b"])
        hid = blk["fc1"](n2)
        # GELU in front of the unfold (a gather with zero padding: GELU commutes with it, 5.4x fewer erf evaluations)
        folded = ops.ffn_fold_gelu(hid, b * t, fh, fw, H, W, 40)
        if isinstance(blk["fc2"], PackedConvX):
            y = torch.empty((x1.shape[0], 512), dtype=torch.float32, device=x1.device)
            blk["fc2"]([folded], out=y.view(b * t, fh, fw, 512), residual=x1.view(b * t, fh, fw, 512))
            return y, x1
        unf = ops.ffn_unfold(folded, fh, fw, out=hid)
        return blk["fc2"](unf, residual=x1), x1

    def compose(self, tokens, enc, b, t, fh, fw):
        """SoftComp + residual with the encoder features (tfocal_transformer.py:65-72, e2fgvi.py:258)."""
        _, h, w, ch = enc.shape
        emb = self.sc(tokens)
        if self.hq:
            folded = ops.softcomp_fold(emb, b * t, fh, fw, h, w, ch)
            return self.sc_bias_conv([folded], residual=enc)
        return ops.softcomp_fold(emb, b * t, fh, fw, h, w, ch, bias_hwc=self.sc_bias_hwc, residual=enc)

    # ------------------------------------------------------------------ decoder
    def decode(self, x):
        n, h, w, _ = x.shape
        lr = dict(act=ACT_LRELU, slope=0.2)
        d = self.dec
        x = ops.resize_bilinear(x, (2 * h, 2 * w), True)
        x = d[0]([x], **lr)
        x = d[1]([x], **lr)
        x = ops.resize_bilinear(x, (4 * h, 4 * w), True)
        x = d[2]([x], **lr)
        return d[3]([x], act=ACT_TANH, out_nchw=True)

    # ------------------------------------------------------------------ whole forward
    def forward(self, frames, l_t, trace=None):
        b, t, c, H, W = frames.shape
        if H % 4 or W % 4:
            raise ValueError("H and W must be multiples of 4")
        h, w = H // 4, W // 4
        fh, fw = token_grid(h, w)
        if fh % 5 or fw % 9:
            raise ValueError("token grid %dx%d is not a multiple of (5,9): pad H to a multiple of 60 and W to a "
                             "multiple of 108 (reference test.py:156-165)" % (fh, fw))
        if not self.hq and (h, w) != (60, 108):
            raise ValueError("model 'e2fgvi' is fixed to 432x240 inputs (sc.bias is [128,60,108]); use e2fgvi_hq")
        if not (1 <= l_t <= t):
            raise ValueError("num_local_frames must be in [1, t]")
        frames = ops._chk(frames.float().contiguous(), "masked_frames")
        if self.bf16:
            return self.forward_x(frames, l_t, b, t, h, w, fh, fw, trace)
        if l_t == 1:
            # a one-frame local window (test.py on a 1-frame video): the reference's flow tensors are empty
            # [b,0,2,h,w] and each propagation direction is backbone(cat(x, 0)) (feat_prop.py:105,131-137)
            fwd = bwd = torch.empty((b, 0, h, w, 2), dtype=torch.float32, device=frames.device)
            enc = self.encode(frames)
        elif self.overlap_flows:
            # SPyNet (many short, small-channel launches) and the encoder (few large launches) are independent:
            # run them on two HIP streams so the flow pyramid fills the gaps of the encoder (fork/join is
            # captured as two branches when the forward is recorded into a HIP graph)
            main = torch.cuda.current_stream()
            side = self._side_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                fwd, bwd = self.flows(frames, l_t)
            enc = self.encode(frames, join=lambda: main.wait_stream(side))
        else:
            fwd, bwd = self.flows(frames, l_t)
            enc = self.encode(frames)
        ch = enc.shape[3]
        if trace is not None:
            trace["flow_fwd"], trace["flow_bwd"], trace["enc"] = fwd, bwd, enc.clone()
        enc5 = enc.view(b, t, h, w, ch)
        if b == 1:
            loc = enc5[0, :l_t].unsqueeze(1)                 # view: [l_t, 1, h, w, C]
            # one clip: the local frames are a view of the encoder output and the propagated features replace them in place (no
            # 33 MB copy; under E2FGVI_AUTOTUNE=1 the fusion layer writes a fresh tensor so that its candidates can be timed)
            inplace = not ops.AUTOTUNE
            prop = self.propagate(loc, fwd, bwd, inplace=inplace)
            if not inplace:
                enc5[0, :l_t].copy_(prop[:, 0])
        else:
            loc = enc5[:, :l_t].permute(1, 0, 2, 3, 4).contiguous()
            prop = self.propagate(loc, fwd, bwd)
            enc5[:, :l_t].copy_(prop.permute(1, 0, 2, 3, 4))
        if trace is not None:
            trace["prop"] = enc.clone()
        tok = self.soft_split(enc).view(b * t * fh * fw, 512)
        if trace is not None:
            trace["tokens0"] = tok.clone()
        for i in range(8):
            tok, x1 = self.block(i, tok, b, t, fh, fw, (h, w))
            if trace is not None:
                trace["block%d_attn_out" % i] = x1
                trace["tokens%d" % (i + 1)] = tok
        dec_in = self.compose(tok, enc, b, t, fh, fw)
        if trace is not None:
            trace["dec_in"] = dec_in
        out = self.decode(dec_in)
        if l_t == 1:
            empty = torch.empty((b, 0, 2, h, w), dtype=torch.float32, device=out.device)
            return out, (empty, empty.clone())
        flows_out = (ops.nhwc_to_nchw(fwd.reshape(b * (l_t - 1), h, w, 2)).view(b, l_t - 1, 2, h, w),
                     ops.nhwc_to_nchw(bwd.reshape(b * (l_t - 1), h, w, 2)).view(b, l_t - 1, 2, h, w))
        return out, flows_out
