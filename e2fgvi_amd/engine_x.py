"""bf16 data path of the forward (BASELINE.json configs 4 / 5: e2fgvi_hq at 720p / 1080p, "bf16 MFMA").

Same stages, same reference call sites and the same orchestration as engine.Engine's fp32 path; what changes is where the
bytes live and which pipe multiplies them:

  * activations are bf16 NHWC tensors in HBM (half the traffic of the fp32 path), every conv / linear runs in
    csrc/conv_bf16x.hip (bf16 operands by LDS-DMA, v_mfma_f32_32x32x16_bf16, fp32 accumulation and epilogue);
  * kept in fp32: the SPyNet flow pyramid and the flows (sub-pixel sampling positions), the DCN offsets / masks (output of
    conv_offset.6 incl. its 10 tanh + flow post-processing), the token residual stream between transformer blocks
    (LayerNorm input) and the output frames; the recurrent propagation features are ONE bf16 tensor per step (conv source,
    flow-warp source, and -- re-laid out [group][pixel][16] -- the deformable conv's gather source);
  * LayerNorm, window pooling, fold / unfold + GELU, SoftComp fold and the x2 upsamples read / write bf16 and compute
    in fp32 (typed variants of the fp32 kernels, csrc/misc.hip).
"""
import torch

from . import ops
from .ops import ACT_DCNPOST, ACT_LRELU, ACT_NONE, ACT_TANH, PackedConvX, PackedLinearX

BF16 = torch.bfloat16
# Settled in rounds 2-5 and no longer switchable (round 6): decoder.6 on csrc/conv_tail.hip; the deformable conv gathers from
# [group][pixel][16] copies of the propagated features; the recurrent propagation state and the flow-warp sources are the bf16
# copies (an fp32 state changes the end-to-end error by < 1 % of itself: tools/bf16_error_growth.py); the FFN's second Linear is
# the 7x7 / stride-3 conv of the folded tensor; SoftComp (HQ) runs in gather form.


class BF16Path:
    # ------------------------------------------------------------------ weights
    def _init_x(self, f):
        dev = self.device
        # encoder: the 3-channel frames are carried as 8 bf16 channels (16-byte pixels), layer 0's weight is padded to match
        w0 = torch.zeros(64, 8, 3, 3, device=dev)
        w0[:, :3] = f("encoder.layers.0.weight")
        enc = [PackedConvX(w0, f("encoder.layers.0.bias"), [8], stride=2, pad=1)]
        for i, (cpg, g, s) in zip((2, 4, 6, 8, 10, 12, 14, 16),
                                  (([64], 1, 1), ([64], 1, 2), ([128], 1, 1), ([256], 1, 1), ([128, 192], 2, 1),
                                   ([64, 128], 4, 1), ([32, 48], 8, 1), ([256, 256], 1, 1))):
            enc.append(PackedConvX(f("encoder.layers.%d.weight" % i), f("encoder.layers.%d.bias" % i), cpg, groups=g, stride=s, pad=1))
        self.xenc = enc
        for k, i in enumerate((0, 2, 4, 6, 8, 10, 12, 14, 16)):
            enc[k].name = "encoder.layers.%d" % i
        self.xdec = [PackedConvX(f("decoder.0.conv.weight"), f("decoder.0.conv.bias"), [128], pad=1),
                     PackedConvX(f("decoder.2.weight"), f("decoder.2.bias"), [128], pad=1),
                     PackedConvX(f("decoder.4.conv.weight"), f("decoder.4.conv.bias"), [64], pad=1),
                     ops.PackedTailConv(f("decoder.6.weight"), f("decoder.6.bias"), dtype=BF16)]
        for k, n in enumerate(("decoder.0.conv", "decoder.2", "decoder.4.conv", "decoder.6")):
            self.xdec[k].name = n
        self.xprop = {}
        for d, nparts in (("backward_", 2), ("forward_", 3)):
            p = "feat_prop_module.deform_align.%s." % d
            w_off0 = f(p + "conv_offset.0.weight")                       # [128, 388]: cond_n1, cur, cond_n2, flows(4)
            w_off0 = torch.cat([w_off0, w_off0.new_zeros(128, 4, 3, 3)], 1)   # the flows arrive as 8 bf16 channels
            off = [PackedConvX(w_off0, f(p + "conv_offset.0.bias"), [128, 128, 128, 8], pad=1),
                   PackedConvX(f(p + "conv_offset.2.weight"), f(p + "conv_offset.2.bias"), [128], pad=1),
                   PackedConvX(f(p + "conv_offset.4.weight"), f(p + "conv_offset.4.bias"), [128], pad=1),
                   PackedConvX(f(p + "conv_offset.6.weight"), f(p + "conv_offset.6.bias"), [128], pad=1)]
            b = "feat_prop_module.backbone.%s." % d
            bb = [PackedConvX(f(b + "0.weight"), f(b + "0.bias"), [128] * nparts, pad=1),
                  PackedConvX(f(b + "2.weight"), f(b + "2.bias"), [128], pad=1)]
            dcn = ops.PackedDcn(f(p + "weight"), f(p + "bias"), 16, pad=1, mfma="bf16")
            dcn.name = "deform_align.%sdcn" % d
            for k, c in enumerate(off):
                c.name = "deform_align.%sconv_offset.%d" % (d, 2 * k)
            bb[0].name, bb[1].name = "backbone.%s0" % d, "backbone.%s2" % d
            self.xprop[d] = (off, bb, dcn)
        self.xfusion = PackedConvX(f("feat_prop_module.fusion.weight"), f("feat_prop_module.fusion.bias"), [128, 128])
        self.xss = PackedConvX(f("ss.embedding.weight").view(512, 128, 7, 7), f("ss.embedding.bias"), [128], stride=3, pad=3)
        wsc = f("sc.embedding.weight").view(128, 49, 512).permute(1, 0, 2).reshape(6272, 512).contiguous()
        bsc = f("sc.embedding.bias").view(128, 49).t().reshape(6272).contiguous()
        self.xsc = PackedLinearX(wsc, bsc)
        self.xfusion.name, self.xss.name, self.xsc.name = "fusion", "ss.embedding", "sc.embedding"
        if self.hq:
            self.xsc_bias_conv = PackedConvX(f("sc.bias_conv.weight"), f("sc.bias_conv.bias"), [128], pad=1)
            self.xsc_bias_conv.name = "sc.bias_conv"
            # SoftComp in gather form (nine phase convolutions writing the folded image directly: no [tokens, 6272] tensor,
            # 813 MB at 720p T=10, and no fold kernel); other token grids than 3 x the feature size take the Linear + fold pair
            self.xsc_gather = ops.SoftCompGather(f("sc.embedding.weight"), f("sc.embedding.bias"), 128)
        # SPyNet: the conv stacks of the six pyramid levels on bf16 MFMA (they are 15 % of the 720p forward in fp32); the
        # geometry stays fp32 -- pyramid images, warps, the flow itself and the residual sum flow = up(flow) + net(...)
        # (the last conv of a level adds the fp32 upsampled flow and stores fp32).
        self.xspy = []
        for lv in range(6):
            convs = []
            for j, cin in enumerate((8, 32, 64, 32, 16)):
                p = "update_spynet.basic_module.%d.basic_module.%d.conv." % (lv, j)
                convs.append(PackedConvX(f(p + "weight"), f(p + "bias"), [cin], pad=3))
                convs[-1].name = "spynet.%d.%d" % (lv, j)
            self.xspy.append(convs)
        self.xblocks = []
        for i in range(8):
            p = "transformer.%d." % i
            w1 = f(p + "mlp.conv1.0.weight").view(40, 49, 512).permute(1, 0, 2).reshape(1960, 512).contiguous()
            b1 = f(p + "mlp.conv1.0.bias").view(40, 49).t().reshape(1960).contiguous()
            w2 = f(p + "mlp.conv2.1.weight").view(512, 40, 49).permute(0, 2, 1).reshape(512, 1960).contiguous()
            blk = dict(qkv=PackedLinearX(f(p + "attn.qkv.weight"), f(p + "attn.qkv.bias")),
                       proj=PackedLinearX(f(p + "attn.proj.weight"), f(p + "attn.proj.bias")),
                       fc1=PackedLinearX(w1, b1),
                       # Linear(1960 -> 512) of the unfolded 7x7 patches == the 7x7 / stride 3 / pad 3 convolution of the folded
                       # [F, H, W, 40] tensor (tfocal_transformer.py:81,95-97): no unfold kernel, no [rows, 1960] tensor
                       fc2=PackedConvX(f(p + "mlp.conv2.1.weight").view(512, 40, 7, 7), f(p + "mlp.conv2.1.bias"), [40],
                                       stride=3, pad=3))
            for k in ("qkv", "proj", "fc1", "fc2"):
                blk[k].name = "transformer.%d.%s" % (i, k)
            self.xblocks.append(blk)

        if self.autotune_x:
            # the best tile of conv_bf16x depends on the layer's M x N x K and on how many tiles that makes for 256 CUs:
            # timed on the first (eager) call of each size class, never under graph capture
            layers = self.xenc + self.xdec + [self.xfusion, self.xss, self.xsc] + [c for cv in self.xspy for c in cv]
            for off, bb, _ in self.xprop.values():
                layers += off + bb
            for blk in self.xblocks:
                layers += [blk[k] for k in ("qkv", "proj", "fc1", "fc2")]
            if self.hq:
                layers.append(self.xsc_bias_conv)
            for c in layers:
                c.tune = True

    def _zero16(self, shape):
        key = ("bf16",) + tuple(shape)
        if key not in self._zeros:
            self._zeros[key] = torch.zeros(tuple(shape), dtype=BF16, device=self.device)
        return self._zeros[key]

    # ------------------------------------------------------------------ flows (e2fgvi.py:210-234, flow_comp.py:84-169)
    def spynet_level_x(self, lv, pyr_lv, ref_idx, supp_idx, flow):
        from .ops import ACT_RELU
        inp, inp16 = ops.spynet_level_input(pyr_lv, ref_idx, supp_idx, flow, bf16_copy=True)
        cv = self.xspy[lv]
        x = cv[0]([inp16], act=ACT_RELU)
        x = cv[1]([x], act=ACT_RELU)
        x = cv[2]([x], act=ACT_RELU)
        x = cv[3]([x], act=ACT_RELU)
        return cv[4]([x], out_dtype=torch.float32, residual=inp, res_coff=6)          # + fp32 upsampled flow

    # ------------------------------------------------------------------ encoder (e2fgvi.py:96-109)
    def encode_x(self, frames):
        b, t, c, H, W = frames.shape
        x = ops.nchw_to_nhwc(frames.reshape(b * t, c, H, W).contiguous(), ld=8, out_dtype=BF16)
        e = self.xenc
        lr = dict(act=ACT_LRELU, slope=0.2)
        x = e[0]([x], **lr)
        x = e[1]([x], **lr)
        x = e[2]([x], **lr)
        x0 = e[3]([x], **lr)
        x = e[4]([x0], **lr)
        for k in (5, 6, 7, 8):
            x = e[k]([x0, x], **lr)
        return x                                            # bf16 [b*t, h, w, 128]

    # ------------------------------------------------------------------ propagation (feat_prop.py:81-149, :35-58)
    def propagate_x(self, loc, flows_a, flows_b):
        """loc: bf16 [l_t, b, h, w, 128]; flows fp32 NHWC [b,l_t-1,h,w,2].  Returns bf16 [l_t, b, h, w, 128]."""
        l_t, b, h, w, ch = loc.shape
        dev = loc.device
        stores = {}
        zero16 = self._zero16((b, h, w, ch))
        lk = dict(act=ACT_LRELU, slope=0.1)
        for name, flows in (("backward_", flows_a), ("forward_", flows_b)):
            off, bb, dcn = self.xprop[name]
            store16 = torch.empty((l_t, b, h, w, ch), dtype=BF16, device=dev)
            order = list(range(l_t))
            if name == "backward_":
                order = order[::-1]
            img_stride = (l_t - 1) * h * w * 2
            hist16 = []                     # bf16 propagated features in processing order: conv, flow-warp and DCN sources
            # ... re-laid out [group][pixel][16]: the 32-byte runs a deform group's samples fetch are then adjacent for
            # neighbouring pixels and share cache lines (NHWC: one run per 256-byte pixel) -- tools/dcn_bench_x.py
            planar16 = []
            zero16p = self._zero16((ch // 16, b, h, w, 16))
            aligned = zero16
            for i, idx in enumerate(order):
                cur = loc[idx]
                if i > 0:
                    flow_a = flows[0, i - 1]
                    flow_b = flows[0, i - 2] if i > 1 else None
                    feat_n2 = hist16[-2] if i > 1 else None
                    cond, fl, fl8 = ops.prop_cond(hist16[-1], feat_n2, flow_a, flow_b, img_stride, cond_dtype=BF16, flows8=True)
                    x = off[0]([(cond, 0), cur, (cond, ch), fl8], **lk)
                    x = off[1]([x], **lk)
                    x = off[2]([x], **lk)
                    offs = off[3]([x], out_dtype=torch.float32, residual=fl, act=ACT_DCNPOST, slope=10.0)
                    aligned = dcn([planar16[-1], planar16[-2] if i > 1 else zero16p], offs, out_dtype=BF16, planar=True)
                srcs = [cur, stores["backward_"][idx], aligned] if name == "forward_" else [cur, aligned]
                y = bb[0](srcs, **lk)
                bb[1]([y], out=store16[idx], residual=aligned)     # one bf16 result: conv, warp and (re-laid out) DCN source of later steps
                hist16.append(store16[idx])
                if i + 1 < l_t:
                    planar16.append(ops.to_planar16(store16[idx]))
            stores[name] = store16
        out = self.xfusion([stores["backward_"].view(l_t * b, h, w, ch), stores["forward_"].view(l_t * b, h, w, ch)],
                           residual=loc.view(l_t * b, h, w, ch))
        return out.view(l_t, b, h, w, ch)

    # ------------------------------------------------------------------ transformer (tfocal_transformer*.py)
    def block_x(self, i, x, b, t, fh, fw, hw, want_bf16_copy=False):
        """x: fp32 [b*t*fh*fw, 512] token residual stream.  Returns (x_out fp32, attention-branch output fp32, bf16 copy or None)."""
        blk, xb = self.blocks[i], self.xblocks[i]
        H, W = hw
        tab, nk = self._table(fh, fw, blk)
        rows = x.shape[0]
        prow = b * t * (fh // 5) * (fw // 9)
        nbuf = torch.empty((rows + prow, 512), dtype=BF16, device=x.device)
        n1 = ops.layernorm(x, blk["n1w"], blk["n1b"], out=nbuf[:rows])
        ops.window_pool(n1, blk["pool_w"], blk["pool_b"], b * t, fh, fw, out=nbuf[rows:])
        both = xb["qkv"](nbuf)                                        # bf16 [rows + prow, 1536]
        att = ops.focal_attention_bf16(both[:rows], both[rows:], tab, nk, b, t, fh, fw)
        x1 = xb["proj"](att, out_dtype=torch.float32, residual=x)
        n2 = ops.layernorm(x1, blk["n2w"], blk["n2b"], out_dtype=BF16)
        hid = xb["fc1"](n2)
        # GELU in front of the unfold (a gather with zero padding: GELU commutes with it, 5.4x fewer erf evaluations)
        folded = ops.ffn_fold_gelu(hid, b * t, fh, fw, H, W, 40)
        copy = torch.empty((rows, 512), dtype=BF16, device=x.device) if want_bf16_copy else None
        y = torch.empty((rows, 512), dtype=torch.float32, device=x.device)
        xb["fc2"]([folded], out=y.view(b * t, fh, fw, 512), residual=x1.view(b * t, fh, fw, 512),
                  out2=None if copy is None else copy.view(b * t, fh, fw, 512))
        return y, x1, copy

    def compose_x(self, tok16, enc, b, t, fh, fw):
        """SoftComp + residual with the encoder features (tfocal_transformer.py:65-72, e2fgvi.py:258), bf16."""
        _, h, w, ch = enc.shape
        if self.hq and self.xsc_gather is not None and (h, w) == (3 * fh, 3 * fw):
            folded = self.xsc_gather(tok16.view(b * t, fh, fw, 512))
            return self.xsc_bias_conv([folded], residual=enc)
        emb = self.xsc(tok16)
        if self.hq:
            folded = ops.softcomp_fold(emb, b * t, fh, fw, h, w, ch)
            return self.xsc_bias_conv([folded], residual=enc)
        return ops.softcomp_fold(emb, b * t, fh, fw, h, w, ch, bias_hwc=self.sc_bias_hwc, residual=enc)

    def decode_x(self, x):
        n, h, w, _ = x.shape
        lr = dict(act=ACT_LRELU, slope=0.2)
        d = self.xdec
        x = ops.resize_bilinear(x, (2 * h, 2 * w), True)
        x = d[0]([x], **lr)
        x = d[1]([x], **lr)
        x = ops.resize_bilinear(x, (4 * h, 4 * w), True)
        x = d[2]([x], **lr)
        return d[3]([x], act=ACT_TANH, out_nchw=True)             # 64 -> 3, tanh, fp32 NCHW frames

    # ------------------------------------------------------------------ whole forward (e2fgvi_hq.py:235-263)
    def forward_x(self, frames, l_t, b, t, h, w, fh, fw, trace=None):
        if l_t == 1:
            fwd = bwd = torch.empty((b, 0, h, w, 2), dtype=torch.float32, device=frames.device)
            enc = self.encode_x(frames)
        elif self.overlap_flows:
            main = torch.cuda.current_stream()
            side = self._side_stream()
            side.wait_stream(main)
            with torch.cuda.stream(side):
                fwd, bwd = self.flows(frames, l_t)
            enc = self.encode_x(frames)
            main.wait_stream(side)
        else:
            fwd, bwd = self.flows(frames, l_t)
            enc = self.encode_x(frames)
        ch = enc.shape[3]
        enc5 = enc.view(b, t, h, w, ch)
        if b == 1:
            loc = enc5[0, :l_t].unsqueeze(1)
            prop = self.propagate_x(loc, fwd, bwd)
            enc5[0, :l_t].copy_(prop[:, 0])
        else:
            loc = enc5[:, :l_t].permute(1, 0, 2, 3, 4).contiguous()
            prop = self.propagate_x(loc, fwd, bwd)
            enc5[:, :l_t].copy_(prop.permute(1, 0, 2, 3, 4))
        if trace is not None:
            trace["flow_fwd"], trace["flow_bwd"], trace["prop"] = fwd, bwd, enc.float()
        tok = self.xss([enc], out_dtype=torch.float32).view(b * t * fh * fw, 512)
        tok16 = None
        for i in range(8):
            tok, x1, tok16 = self.block_x(i, tok, b, t, fh, fw, (h, w), want_bf16_copy=(i == 7))
            if trace is not None:
                trace["tokens%d" % (i + 1)] = tok
        dec_in = self.compose_x(tok16, enc, b, t, fh, fw)
        out = self.decode_x(dec_in)
        if l_t == 1:
            empty = torch.empty((b, 0, 2, h, w), dtype=torch.float32, device=out.device)
            return out, (empty, empty.clone())
        flows_out = (ops.nhwc_to_nchw(fwd.reshape(b * (l_t - 1), h, w, 2)).view(b, l_t - 1, 2, h, w),
                     ops.nhwc_to_nchw(bwd.reshape(b * (l_t - 1), h, w, 2)).view(b, l_t - 1, 2, h, w))
        return out, flows_out
