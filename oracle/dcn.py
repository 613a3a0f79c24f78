"""CPU restatement of mmcv-full 1.4.8 ``modulated_deform_conv2d`` (TEST INFRASTRUCTURE).

The op is third-party to the reference (environment.yml:135, README.md:104); its source
is NOT under /root/reference and mmcv is not installed here, so this file restates the
published algorithm of ``modulated_deformable_im2col`` + GEMM:

  for column element (c, tap k=(i,j), out pixel (y,x)), g = c // (C/dg):
      dy = offset[n, g*2*K + 2k,   y, x]      dx = offset[n, g*2*K + 2k+1, y, x]
      m  = mask  [n, g*K + k, y, x]
      py = y*stride - pad + i*dil + dy        px = x*stride - pad + j*dil + dx
      val = 0 unless (py > -1 and px > -1 and py < H and px < W); otherwise bilinear with
            floor(), each of the 4 corners contributing only if inside [0,H-1]x[0,W-1]
      col = val * m
  out = W[Co, C*K] @ col + bias

PARITY UNPINNED for this one op: mmcv-full 1.4.8 is neither under /root/reference nor installable here, and the
reference has no tests or golden vectors at this boundary, so the restatement cannot be checked against mmcv's own
output.  Everything else in the oracle is pinned against the reference's own Python run in the build container.
Parity anchors (no golden vectors exist upstream -- SURVEY.md 8c): the reference call site
model/modules/feat_prop.py:55-58, and the cross-checks in tests/test_oracle_dcn.py
(zero offset + unit mask == F.conv2d; integer offsets == shifted conv; constant sub-pixel
offsets == F.grid_sample(zeros, align_corners=True); second independent restatement in C,
oracle/dcn_ref.c).
"""
import torch


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


def deform_columns(x, offset, mask, kernel_size, stride, padding, dilation, deform_groups):
    """Return the modulated, bilinearly sampled im2col tensor [N, C, K, Ho*Wo]."""
    N, C, H, W = x.shape
    kh, kw = kernel_size
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    dh, dw = _pair(dilation)
    K = kh * kw
    Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) // sh + 1
    Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) // sw + 1
    dg = deform_groups
    cg = C // dg
    assert offset.shape == (N, dg * 2 * K, Ho, Wo), offset.shape
    assert mask.shape == (N, dg * K, Ho, Wo), mask.shape

    off = offset.reshape(N, dg, K, 2, Ho, Wo)
    dy, dx = off[:, :, :, 0], off[:, :, :, 1]                     # [N,dg,K,Ho,Wo]
    m = mask.reshape(N, dg, K, Ho, Wo)

    ys = torch.arange(Ho, dtype=x.dtype).view(1, 1, 1, Ho, 1) * sh - ph
    xs = torch.arange(Wo, dtype=x.dtype).view(1, 1, 1, 1, Wo) * sw - pw
    ki = (torch.arange(K) // kw).to(x.dtype).view(1, 1, K, 1, 1) * dh
    kj = (torch.arange(K) % kw).to(x.dtype).view(1, 1, K, 1, 1) * dw
    py = ys + ki + dy
    px = xs + kj + dx

    inside = (py > -1) & (px > -1) & (py < H) & (px < W)
    y0 = torch.floor(py)
    x0 = torch.floor(px)
    ly, lx = py - y0, px - x0
    hy, hx = 1 - ly, 1 - lx
    y0 = y0.long()
    x0 = x0.long()
    y1, x1 = y0 + 1, x0 + 1

    xf = x.reshape(N, dg, cg, H * W)

    def corner(yy, xx, wgt):
        ok = inside & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
        idx = (yy.clamp(0, H - 1) * W + xx.clamp(0, W - 1)).reshape(N, dg, 1, K * Ho * Wo)
        v = torch.gather(xf, 3, idx.expand(N, dg, cg, K * Ho * Wo))
        wgt = (wgt * ok.to(x.dtype)).reshape(N, dg, 1, K * Ho * Wo)
        return v * wgt

    val = (corner(y0, x0, hy * hx) + corner(y0, x1, hy * lx) +
           corner(y1, x0, ly * hx) + corner(y1, x1, ly * lx))
    val = val * m.reshape(N, dg, 1, K * Ho * Wo)
    return val.reshape(N, C, K, Ho * Wo), (Ho, Wo)


def modulated_deform_conv2d(x, offset, mask, weight, bias=None, stride=1, padding=0,
                            dilation=1, groups=1, deform_groups=1):
    """Same signature as mmcv.ops.modulated_deform_conv2d (call site feat_prop.py:55-58)."""
    N, C, H, W = x.shape
    Co, Cig, kh, kw = weight.shape
    cols, (Ho, Wo) = deform_columns(x.float(), offset.float(), mask.float(), (kh, kw),
                                    stride, padding, dilation, deform_groups)
    K = kh * kw
    cols = cols.reshape(N, groups, Cig * K, Ho * Wo)
    w = weight.reshape(groups, Co // groups, Cig * K)
    out = torch.einsum("gok,ngkp->ngop", w, cols).reshape(N, Co, Ho, Wo)
    if bias is not None:
        out = out + bias.view(1, -1, 1, 1)
    return out
