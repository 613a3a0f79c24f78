"""MI355X stand-ins for the two mmcv symbols the reference imports (model/modules/feat_prop.py:7):
``modulated_deform_conv2d`` and ``ModulatedDeformConv2d``, same signatures / argument meaning as
mmcv-full 1.4.8, NCHW torch tensors in and out, computed by the HIP kernel behind
``e2fgvi_mdcn_nhwc`` (include/e2fgvi_hip.h).  Inference only."""
import math

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
