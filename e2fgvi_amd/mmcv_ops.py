"""MI355X stand-ins for every mmcv symbol the reference imports -- the inner operator boundary of SURVEY.md 8(b):

    mmcv.ops.modulated_deform_conv2d / ModulatedDeformConv2d     model/modules/feat_prop.py:7,13,55-58
    mmcv.cnn.ConvModule                                          model/modules/flow_comp.py:7,181-215
    mmcv.cnn.constant_init                                       model/modules/feat_prop.py:8,33
    mmcv.runner.load_checkpoint                                  model/modules/flow_comp.py:8,72

Same signatures / argument meaning as mmcv-full 1.4.8, NCHW torch tensors in and out; the arithmetic runs in the HIP
kernels behind ``e2fgvi_mdcn_nhwc`` / ``e2fgvi_conv2d_nhwc`` (include/e2fgvi_hip.h).  With these, INTEGRATION.md's
"keep the reference's Python, replace only mmcv" route needs nothing from ``oracle/``.  Inference only."""
import math
import os

import torch
import torch.nn as nn

from . import ops

_CACHE = {}


def _packed(weight, bias, deform_groups, stride, padding, dilation):
    key = (weight.data_ptr(), weight._version, None if bias is None else (bias.data_ptr(), bias._version),
           deform_groups, stride, padding, dilation)
    if key not in _CACHE:
        if len(_CACHE) > 16:
            _CACHE.clear()
        _CACHE[key] = ops.PackedDcn(weight, bias, deform_groups, stride, padding, dilation)
    return _CACHE[key]


def _one(v):
    if isinstance(v, (tuple, list)):
        if v[0] != v[1]:
            raise NotImplementedError("anisotropic stride/padding/dilation")
        return int(v[0])
    return int(v)


def modulated_deform_conv2d(input, offset, mask, weight, bias=None, stride=1, padding=0, dilation=1, groups=1,
                            deform_groups=1):
    if groups != 1:
        raise NotImplementedError("groups != 1 (the reference only uses groups=1, feat_prop.py:55-58)")
    layer = _packed(weight, bias, deform_groups, _one(stride), _one(padding), _one(dilation))
    with torch.no_grad():
        x = ops.nchw_to_nhwc(input.float().contiguous())
        off = ops.nchw_to_nhwc(offset.float().contiguous())
        msk = ops.nchw_to_nhwc(mask.float().contiguous())
        return ops.nhwc_to_nchw(layer([x], off, mask=msk))


class ModulatedDeformConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deform_groups=1, bias=True):
        super().__init__()
        k = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, k
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deform_groups = groups, deform_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *k))
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        stdv = 1.0 / math.sqrt(in_channels * k[0] * k[1])
        self.weight.data.uniform_(-stdv, stdv)

    def forward(self, x, offset, mask):
        return modulated_deform_conv2d(x, offset, mask, self.weight, self.bias, self.stride, self.padding,
                                       self.dilation, self.groups, self.deform_groups)


# ----------------------------------------------------------------------------- mmcv.cnn / mmcv.runner
def constant_init(module, val, bias=0):
    """mmcv.cnn.constant_init (used at feat_prop.py:33 to zero conv_offset[-1])."""
    if getattr(module, "weight", None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


class ConvModule(nn.Module):
    """mmcv.cnn.ConvModule as SPyNet uses it (flow_comp.py:181-215): ``conv`` (nn.Conv2d parameters, mmcv's default
    kaiming-normal fan_out init, zero bias) + optional ``activate`` (ReLU); child names matter for the checkpoint keys
    ``...basic_module.N.conv.weight``.  forward runs the HIP implicit-GEMM / halo conv kernel with the ReLU fused."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1, bias="auto",
                 conv_cfg=None, norm_cfg=None, act_cfg=dict(type="ReLU"), inplace=True, **kwargs):
        super().__init__()
        if norm_cfg is not None or conv_cfg is not None or dilation != 1 or kwargs:
            raise NotImplementedError("ConvModule stand-in: only conv (+ReLU), as the reference uses it")
        if act_cfg is not None and act_cfg.get("type") != "ReLU":
            raise NotImplementedError("ConvModule stand-in: act_cfg must be None or ReLU")
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, groups=groups,
                              bias=(bias is True or bias == "auto"))
        self.with_activation = act_cfg is not None
        if self.with_activation:
            self.activate = nn.ReLU(inplace=inplace)
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode="fan_out", nonlinearity="relu")
        if self.conv.bias is not None:
            nn.init.constant_(self.conv.bias, 0)
        self._packed = None

    def forward(self, x):
        c = self.conv
        key = (c.weight.data_ptr(), c.weight._version, None if c.bias is None else (c.bias.data_ptr(), c.bias._version))
        if self._packed is None or self._packed[0] != key:
            cin_g = c.in_channels // c.groups
            pad_c = (-cin_g) % 4                                  # NHWC sources are addressed in 16-byte units
            if pad_c and c.groups != 1:
                raise NotImplementedError("ConvModule stand-in: grouped conv needs in_channels/groups % 4 == 0")
            w = c.weight.detach().float()
            if pad_c:
                w = torch.cat([w, w.new_zeros(w.shape[0], pad_c, *w.shape[2:])], 1)
            layer = ops.PackedConv(w.contiguous(), None if c.bias is None else c.bias.detach().float(), [cin_g + pad_c],
                                   groups=c.groups, stride=_one(c.stride), pad=_one(c.padding))
            self._packed = (key, layer, cin_g + pad_c if c.groups == 1 else c.in_channels)
        _, layer, ld = self._packed
        with torch.no_grad():
            xh = ops.nchw_to_nhwc(x.float().contiguous(), ld=ld)
            y = layer([xh], act=ops.ACT_RELU if self.with_activation else ops.ACT_NONE)
            return ops.nhwc_to_nchw(y)


def load_checkpoint(model, filename, map_location="cpu", strict=False, logger=None):
    """mmcv.runner.load_checkpoint for LOCAL files.  flow_comp.py:59-72 passes the download URL of the pretrained
    SPyNet by default; without network access a URL (or a missing file) is skipped with a message instead of
    failing -- SPyNet's weights then come with the generator checkpoint (``update_spynet.*`` keys)."""
    if isinstance(filename, str) and os.path.isfile(filename):
        sd = torch.load(filename, map_location=map_location)
        if isinstance(sd, dict) and isinstance(sd.get("state_dict"), dict):
            sd = sd["state_dict"]
        sd = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in sd.items()}
        model.load_state_dict(sd, strict=strict)
        return sd
    print("load_checkpoint: %r is not a local file -- skipped (no network access)" % (filename,))
    return None
