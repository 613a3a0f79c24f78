"""CPU restatement ("port") of the reference E2FGVI / E2FGVI-HQ inference forward.

TEST INFRASTRUCTURE ONLY -- the checker, never the product.  Plain PyTorch fp32 on CPU,
written as stateless functions over a ``state_dict`` so that it can travel to the GPU box
(where /root/reference does not exist).  It is pinned against the reference's own Python
(imported read-only via oracle/ref_import.py) by tests/test_oracle_pin.py and against the
golden fixtures in tests/golden/ (made by tests/golden/make_golden.py from the real
reference).  Every function cites the reference lines it restates.

Each stage also returns/records its intermediate so per-stage parity tests can feed the
HIP path with the oracle's *input* of that stage.
"""
import math

import torch
import torch.nn.functional as F

from oracle.dcn import modulated_deform_conv2d

WIN = (5, 9)          # window_size / focal_window, e2fgvi.py:184-185
HEADS = 4             # e2fgvi.py:183
T2T = dict(kernel_size=(7, 7), stride=(3, 3), padding=(3, 3))   # e2fgvi.py:152-155


# --------------------------------------------------------------------------- flow
def flow_warp(x, flow_nhw2, padding_mode="zeros"):
    """flow_comp.py:345-383: sample x at (x+u, y+v); align_corners=True grid_sample."""
    n, c, h, w = x.shape
    gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
    grid = torch.stack((gx, gy), 2).to(x.dtype) + flow_nhw2
    nx = 2.0 * grid[..., 0] / max(w - 1, 1) - 1.0
    ny = 2.0 * grid[..., 1] / max(h - 1, 1) - 1.0
    return F.grid_sample(x, torch.stack((nx, ny), 3), mode="bilinear",
                         padding_mode=padding_mode, align_corners=True)


def spynet_basic(sd, prefix, level, x):
    """flow_comp.py:172-226: five 7x7 convs 8-32-64-32-16-2, ReLU after the first four."""
    for j in range(5):
        p = "%sbasic_module.%d.basic_module.%d.conv." % (prefix, level, j)
        x = F.conv2d(x, sd[p + "weight"], sd[p + "bias"], padding=3)
        if j < 4:
            x = F.relu(x)
    return x


def spynet(sd, prefix, ref, supp, trace=None):
    """flow_comp.py:136-169 (forward) and :84-134 (compute_flow)."""
    h, w = ref.shape[2:]
    w_up = w if w % 32 == 0 else 32 * (w // 32 + 1)
    h_up = h if h % 32 == 0 else 32 * (h // 32 + 1)
    ref = F.interpolate(ref, size=(h_up, w_up), mode="bilinear", align_corners=False)
    supp = F.interpolate(supp, size=(h_up, w_up), mode="bilinear", align_corners=False)
    mean, std = sd[prefix + "mean"], sd[prefix + "std"]
    refs = [(ref - mean) / std]
    supps = [(supp - mean) / std]
    for _ in range(5):
        refs.append(F.avg_pool2d(refs[-1], 2, 2, count_include_pad=False))
        supps.append(F.avg_pool2d(supps[-1], 2, 2, count_include_pad=False))
    refs, supps = refs[::-1], supps[::-1]
    n = ref.shape[0]
    flow = ref.new_zeros(n, 2, h_up // 32, w_up // 32)
    for level in range(6):
        if level == 0:
            flow_up = flow
        else:
            flow_up = F.interpolate(flow, scale_factor=2, mode="bilinear", align_corners=True) * 2.0
        warped = flow_warp(supps[level], flow_up.permute(0, 2, 3, 1), padding_mode="border")
        flow = flow_up + spynet_basic(sd, prefix, level, torch.cat([refs[level], warped, flow_up], 1))
        if trace is not None:
            trace["spynet_level%d" % level] = flow
    flow = F.interpolate(flow, size=(h, w), mode="bilinear", align_corners=False)
    flow = flow.clone()
    flow[:, 0] *= float(w) / float(w_up)
    flow[:, 1] *= float(h) / float(h_up)
    return flow


def bidirect_flow(sd, local_frames01, trace=None):
    """e2fgvi.py:210-234.  local_frames01: [b,l_t,3,H,W] in [0,1]."""
    b, l_t, c, h, w = local_frames01.shape
    small = F.interpolate(local_frames01.reshape(-1, c, h, w), scale_factor=1 / 4, mode="bilinear",
                          align_corners=True, recompute_scale_factor=True)
    small = small.view(b, l_t, c, h // 4, w // 4)
    f1 = small[:, :-1].reshape(-1, c, h // 4, w // 4)
    f2 = small[:, 1:].reshape(-1, c, h // 4, w // 4)
    fwd = spynet(sd, "update_spynet.", f1, f2, trace).view(b, l_t - 1, 2, h // 4, w // 4)
    bwd = spynet(sd, "update_spynet.", f2, f1).view(b, l_t - 1, 2, h // 4, w // 4)
    return fwd, bwd


# --------------------------------------------------------------------------- encoder / decoder
def encoder(sd, x):
    """e2fgvi.py:71-109 (HQ: e2fgvi_hq.py:96-110): 9 convs + LeakyReLU(0.2); layers 10..16 see a
    per-group interleaved concat of x0 (output of layer 6's activation) and the running output."""
    bt = x.shape[0]
    strides = {0: 2, 4: 2}
    groups = {10: 2, 12: 4, 14: 8, 16: 1}
    out = x
    x0 = None
    for i in range(0, 18, 2):
        if i == 8:
            x0 = out
        if i > 8:
            g = groups[i]
            h, w = x0.shape[2:]
            out = torch.cat([x0.view(bt, g, -1, h, w), out.view(bt, g, -1, h, w)], 2).view(bt, -1, h, w)
        p = "encoder.layers.%d." % i
        out = F.conv2d(out, sd[p + "weight"], sd[p + "bias"], stride=strides.get(i, 1), padding=1,
                       groups=groups.get(i, 1))
        out = F.leaky_relu(out, 0.2)
    return out


def decoder(sd, x):
    """e2fgvi.py:112-130, 143-150, 261-262."""
    def up(t):
        return F.interpolate(t, scale_factor=2, mode="bilinear", align_corners=True)
    x = F.leaky_relu(F.conv2d(up(x), sd["decoder.0.conv.weight"], sd["decoder.0.conv.bias"], padding=1), 0.2)
    x = F.leaky_relu(F.conv2d(x, sd["decoder.2.weight"], sd["decoder.2.bias"], padding=1), 0.2)
    x = F.leaky_relu(F.conv2d(up(x), sd["decoder.4.conv.weight"], sd["decoder.4.conv.bias"], padding=1), 0.2)
    x = F.conv2d(x, sd["decoder.6.weight"], sd["decoder.6.bias"], padding=1)
    return torch.tanh(x)


# --------------------------------------------------------------------------- propagation
def deform_align(sd, prefix, x, extra_feat, flow_1, flow_2, trace=None):
    """feat_prop.py:35-58 (SecondOrderDeformableAlignment.forward)."""
    t = torch.cat([extra_feat, flow_1, flow_2], 1)
    for j in (0, 2, 4, 6):
        t = F.conv2d(t, sd["%sconv_offset.%d.weight" % (prefix, j)], sd["%sconv_offset.%d.bias" % (prefix, j)], padding=1)
        if j < 6:
            t = F.leaky_relu(t, 0.1)
    o1, o2, mask = torch.chunk(t, 3, dim=1)
    offset = 10.0 * torch.tanh(torch.cat((o1, o2), 1))          # max_residue_magnitude = 10
    off1, off2 = torch.chunk(offset, 2, dim=1)
    off1 = off1 + flow_1.flip(1).repeat(1, off1.size(1) // 2, 1, 1)
    off2 = off2 + flow_2.flip(1).repeat(1, off2.size(1) // 2, 1, 1)
    offset = torch.cat([off1, off2], 1)
    mask = torch.sigmoid(mask)
    if trace is not None:
        trace.setdefault("dcn_calls", []).append(dict(x=x, offset=offset, mask=mask, raw=t))
    return modulated_deform_conv2d(x, offset, mask, sd[prefix + "weight"], sd[prefix + "bias"],
                                   1, 1, 1, 1, 16)


def propagate(sd, x, flows_backward, flows_forward, trace=None):
    """feat_prop.py:81-149.  NOTE the positional binding at e2fgvi.py:249-250: the caller passes
    (pred_flows_forward, pred_flows_backward) into (flows_backward, flows_forward)."""
    b, t, c, h, w = x.shape
    P = "feat_prop_module."
    spatial = [x[:, i] for i in range(t)]
    feats = {}
    for name in ("backward_", "forward_"):
        feats[name] = []
        frame_idx = list(range(t))
        flow_idx = list(range(-1, t - 1))              # NOT reversed for backward (feat_prop.py:94-103)
        if name == "backward_":
            frame_idx = frame_idx[::-1]
            flows = flows_backward
        else:
            flows = flows_forward
        feat_prop = x.new_zeros(b, c, h, w)
        for i, idx in enumerate(frame_idx):
            cur = spatial[idx]
            if i > 0:
                flow_n1 = flows[:, flow_idx[i]]
                cond_n1 = flow_warp(feat_prop, flow_n1.permute(0, 2, 3, 1))
                feat_n2 = torch.zeros_like(feat_prop)
                flow_n2 = torch.zeros_like(flow_n1)
                cond_n2 = torch.zeros_like(cond_n1)
                if i > 1:
                    feat_n2 = feats[name][-2]
                    flow_n2 = flows[:, flow_idx[i - 1]]
                    flow_n2 = flow_n1 + flow_warp(flow_n2, flow_n1.permute(0, 2, 3, 1))
                    cond_n2 = flow_warp(feat_n2, flow_n2.permute(0, 2, 3, 1))
                cond = torch.cat([cond_n1, cur, cond_n2], 1)
                feat_prop = deform_align(sd, P + "deform_align." + name + ".", torch.cat([feat_prop, feat_n2], 1),
                                         cond, flow_n1, flow_n2, trace)
            parts = [cur]
            if name == "forward_":
                parts.append(feats["backward_"][idx])
            parts.append(feat_prop)
            y = torch.cat(parts, 1)
            bp = P + "backbone." + name + "."
            y = F.leaky_relu(F.conv2d(y, sd[bp + "0.weight"], sd[bp + "0.bias"], padding=1), 0.1)
            y = F.conv2d(y, sd[bp + "2.weight"], sd[bp + "2.bias"], padding=1)
            feat_prop = feat_prop + y
            feats[name].append(feat_prop)
        if name == "backward_":
            feats[name] = feats[name][::-1]
    outs = []
    for i in range(t):
        a = torch.cat([feats["backward_"][i], feats["forward_"][i]], 1)
        outs.append(F.conv2d(a, sd[P + "fusion.weight"], sd[P + "fusion.bias"]))
    return torch.stack(outs, 1) + x


# --------------------------------------------------------------------------- transformer
def token_grid(h, w):
    k, s, p = T2T["kernel_size"], T2T["stride"], T2T["padding"]
    return (int((h + 2 * p[0] - (k[0] - 1) - 1) / s[0] + 1),
            int((w + 2 * p[1] - (k[1] - 1) - 1) / s[1] + 1))


def soft_split(sd, x, b):
    """tfocal_transformer.py:39-46 / _hq.py:32-46."""
    f_h, f_w = token_grid(x.shape[2], x.shape[3])
    feat = F.unfold(x, **T2T).permute(0, 2, 1)
    feat = F.linear(feat, sd["ss.embedding.weight"], sd["ss.embedding.bias"])
    return feat.view(b, -1, f_h, f_w, feat.size(2))


def soft_comp(sd, x, t, out_hw, hq):
    """tfocal_transformer.py:65-72 / _hq.py:67-79."""
    b_, _, _, _, c_ = x.shape
    feat = F.linear(x.view(b_, -1, c_), sd["sc.embedding.weight"], sd["sc.embedding.bias"])
    b, _, c = feat.shape
    feat = feat.view(b * t, -1, c).permute(0, 2, 1)
    feat = F.fold(feat, output_size=out_hw, **T2T)
    if hq:
        return F.conv2d(feat, sd["sc.bias_conv.weight"], sd["sc.bias_conv.bias"], padding=1)
    return feat + sd["sc.bias"][None]


def fusion_ffn(sd, p, x, out_hw):
    """tfocal_transformer.py:89-98 / _hq.py:92-119."""
    x = F.linear(x, sd[p + "conv1.0.weight"], sd[p + "conv1.0.bias"])
    b, n, c = x.shape
    f_h, f_w = token_grid(*out_hw)
    n_vecs = f_h * f_w
    ones = x.new_ones(b, n, 49).view(-1, n_vecs, 49).permute(0, 2, 1)
    normalizer = F.fold(ones, output_size=out_hw, **T2T)
    y = F.fold(x.view(-1, n_vecs, c).permute(0, 2, 1), output_size=out_hw, **T2T)
    y = F.unfold(y / normalizer, **T2T).permute(0, 2, 1).contiguous().view(b, n, c)
    y = F.gelu(y)
    return F.linear(y, sd[p + "conv2.1.weight"], sd[p + "conv2.1.bias"])


def _win_part(x, ws):
    """tfocal_transformer.py:101-114."""
    B, T, H, W, C = x.shape
    x = x.view(B, T, H // ws[0], ws[0], W // ws[1], ws[1], C)
    return x.permute(0, 2, 4, 1, 3, 5, 6).contiguous().view(-1, T * ws[0] * ws[1], C)


def rolled_valid_index():
    """tfocal_transformer.py:169-180 (expand_size = (2,4))."""
    ws, ex = WIN, (WIN[0] // 2, WIN[1] // 2)
    m_tl = torch.ones(ws); m_tl[:-ex[0], :-ex[1]] = 0
    m_tr = torch.ones(ws); m_tr[:-ex[0], ex[1]:] = 0
    m_bl = torch.ones(ws); m_bl[ex[0]:, :-ex[1]] = 0
    m_br = torch.ones(ws); m_br[ex[0]:, ex[1]:] = 0
    return torch.stack((m_tl, m_tr, m_bl, m_br), 0).flatten(0).nonzero(as_tuple=False).view(-1)


def window_attention(sd, p, x, x_pooled, preproj=False, qkv_rows=None, qkv_pool_rows=None):
    """tfocal_transformer.py:210-399.  x: [B,T,H,W,C] (already LayerNorm'ed);
    x_pooled: [B,nWh,nWw,T,C].  ``preproj`` returns the attention output before self.proj
    ([B*nWin, T*45, C], window-major) for kernel-level tests.  ``qkv_rows`` [B,T,H,W,3C] / ``qkv_pool_rows``
    [B,T,nWh,nWw,3C]: use these qkv Linear outputs instead of applying the Linear (tests of the bf16 attention kernel feed
    the bf16-rounded rows the kernel reads)."""
    B, T, nH, nW, C = x.shape
    ws, ex, nh = WIN, (WIN[0] // 2, WIN[1] // 2), HEADS
    hd = C // nh
    qw, qb = (sd[p + "qkv.weight"], sd[p + "qkv.bias"]) if qkv_rows is None else (None, None)
    qkv = (F.linear(x, qw, qb) if qkv_rows is None else qkv_rows).reshape(B, T, nH, nW, 3, C).permute(4, 0, 1, 2, 3, 5).contiguous()
    q, k, v = qkv[0], qkv[1], qkv[2]

    def heads(t):   # -> [B*nWin, nh, T*45, hd]
        return _win_part(t, ws).view(-1, T, ws[0] * ws[1], nh, hd).permute(0, 3, 1, 2, 4).contiguous() \
            .view(-1, nh, T * ws[0] * ws[1], hd)
    q_w, k_w, v_w = heads(q), heads(k), heads(v)

    valid = rolled_valid_index()

    def rolled(t):
        parts = []
        for sy, sx in ((-ex[0], -ex[1]), (-ex[0], ex[1]), (ex[0], -ex[1]), (ex[0], ex[1])):
            r = torch.roll(t, shifts=(sy, sx), dims=(2, 3))
            parts.append(_win_part(r, ws).view(-1, T, ws[0] * ws[1], nh, hd))
        r = torch.cat(parts, 2).permute(0, 3, 1, 2, 4).contiguous()[:, :, :, valid]
        return r.view(-1, nh, T * r.shape[3], hd)
    k_r = torch.cat((k_w, rolled(k)), 2)
    v_r = torch.cat((v_w, rolled(v)), 2)

    # pooled (focal level 1): unfold (5,9), pad (2,4) over the window grid
    xp = x_pooled.permute(0, 3, 1, 2, 4).contiguous()             # B,T,nWh,nWw,C
    nWh, nWw = xp.shape[2:4]
    ones = xp.new_ones(T, 1, nWh, nWw)
    um = F.unfold(ones, kernel_size=ws, padding=(ws[0] // 2, ws[1] // 2)) \
        .view(1, T, ws[0], ws[1], -1).permute(4, 1, 2, 3, 0).contiguous().view(nWh * nWw, -1, 1)
    pmask = um.flatten(1).unsqueeze(0)
    pmask = pmask.masked_fill(pmask == 0, -100.0).masked_fill(pmask > 0, 0.0)   # [1,nWin,T*45]
    qkv_p = (F.linear(xp, qw, qb) if qkv_pool_rows is None else qkv_pool_rows).reshape(B, T, nWh, nWw, 3, C).permute(4, 0, 1, 5, 2, 3) \
        .reshape(3, -1, C, nWh, nWw).contiguous()

    def pooled(t):
        u = F.unfold(t, kernel_size=ws, padding=(ws[0] // 2, ws[1] // 2)) \
            .view(B, T, C, ws[0], ws[1], -1).permute(0, 5, 1, 3, 4, 2).contiguous() \
            .view(-1, T, ws[0] * ws[1], nh, hd).permute(0, 3, 1, 2, 4).contiguous()
        return u.view(-1, nh, T * ws[0] * ws[1], hd)
    k_all = torch.cat([k_r, pooled(qkv_p[1])], 2)
    v_all = torch.cat([v_r, pooled(qkv_p[2])], 2)

    attn = (q_w * hd ** -0.5) @ k_all.transpose(-2, -1)
    wa = T * ws[0] * ws[1]
    off = k_r.shape[2]
    add = pmask[:, :, None, None, :].repeat(attn.shape[0] // pmask.shape[1], 1, 1, 1, 1) \
        .view(-1, 1, 1, pmask.shape[-1])
    attn[:, :, :wa, off:off + wa] = attn[:, :, :wa, off:off + wa] + add
    attn = attn.softmax(-1)
    out = (attn @ v_all).transpose(1, 2).reshape(attn.shape[0], wa, C)
    if preproj:
        return out
    return F.linear(out, sd[p + "proj.weight"], sd[p + "proj.bias"])


def window_reverse(a, B, T, H, W):
    """tfocal_transformer.py:132-147 for [B*nWin, T*45, C] window-major tokens."""
    ws = WIN
    a = a.view(-1, T, ws[0], ws[1], a.shape[-1])
    return a.view(B, H // ws[0], W // ws[1], T, ws[0], ws[1], -1).permute(0, 3, 1, 4, 2, 5, 6).contiguous() \
        .view(B, T, H, W, -1)


def pool_windows(sd, p, xn):
    """tfocal_transformer.py:508-516: Linear(45->1) over each window's tokens -> [B,nWh,nWw,T,C]."""
    B, T, H, W, C = xn.shape
    ws = WIN
    xw = xn.view(B, T, H // ws[0], ws[0], W // ws[1], ws[1], C).permute(0, 2, 4, 1, 3, 5, 6).contiguous()
    nWh, nWw = xw.shape[1:3]
    xw = xw.view(B, nWh, nWw, T, ws[0] * ws[1], C).transpose(4, 5)
    return F.linear(xw, sd[p + "pool_layers.0.weight"], sd[p + "pool_layers.0.bias"]).flatten(-2)


def transformer_block(sd, i, x, out_hw, trace=None):
    """tfocal_transformer.py:466-536 / _hq.py:492-565 (token grid is a multiple of (5,9), so the
    pad/trim branches :490-506 are dead)."""
    p = "transformer.%d." % i
    B, T, H, W, C = x.shape
    ws = WIN
    assert H % ws[0] == 0 and W % ws[1] == 0
    shortcut = x
    xn = F.layer_norm(x, (C,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-5)
    x_pooled = pool_windows(sd, p, xn)
    a = window_attention(sd, p + "attn.", xn, x_pooled)
    x = shortcut + window_reverse(a, B, T, H, W)
    if trace is not None:
        trace["block%d_attn_out" % i] = x
    y = F.layer_norm(x, (C,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-5)
    x = x + fusion_ffn(sd, p + "mlp.", y.view(B, T * H * W, C), out_hw).view(B, T, H, W, C)
    return x


# --------------------------------------------------------------------------- whole forward
def forward(sd, masked_frames, num_local_frames, model="e2fgvi", trace=None):
    """e2fgvi.py:236-263 / e2fgvi_hq.py:235-263.  Returns (frames[b*t,3,H,W], (flow_fwd, flow_bwd))."""
    hq = model == "e2fgvi_hq"
    sd = {k: v.detach().float() for k, v in sd.items() if v.is_floating_point()}
    l_t = num_local_frames
    b, t, c0, H, W = masked_frames.shape
    x = masked_frames.float()
    with torch.no_grad():
        flows = bidirect_flow(sd, (x[:, :l_t] + 1) / 2, trace)
        enc = encoder(sd, x.view(b * t, c0, H, W))
        _, c, h, w = enc.shape
        if trace is not None:
            trace["flow_fwd"], trace["flow_bwd"], trace["enc"] = flows[0], flows[1], enc
        local = enc.view(b, t, c, h, w)[:, :l_t]
        ref = enc.view(b, t, c, h, w)[:, l_t:]
        local = propagate(sd, local, flows[0], flows[1], trace)
        enc = torch.cat((local, ref), 1)
        if trace is not None:
            trace["prop"] = enc
        tok = soft_split(sd, enc.view(-1, c, h, w), b)
        if trace is not None:
            trace["tokens0"] = tok
        for i in range(8):
            tok = transformer_block(sd, i, tok, (h, w), trace)
            if trace is not None:
                trace["tokens%d" % (i + 1)] = tok
        tr = soft_comp(sd, tok, t, (h, w), hq).view(b, t, -1, h, w)
        enc = enc + tr
        if trace is not None:
            trace["dec_in"] = enc
        out = decoder(sd, enc.view(b * t, c, h, w))
    return out, flows
