"""mmcv.ops stand-in (test infrastructure).  Used at feat_prop.py:7,13,55-58."""
import math
import torch
import torch.nn as nn

from oracle.dcn import modulated_deform_conv2d  # noqa: F401  (CPU restatement)


class ModulatedDeformConv2d(nn.Module):
    """Parameter holder with mmcv's attribute names and init (uniform +-1/sqrt(C*kh*kw),
    zero bias).  The reference subclass overrides forward (feat_prop.py:35)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 dilation=1, groups=1, deform_groups=1, bias=True):
        super().__init__()
        k = (kernel_size, kernel_size) if isinstance(kernel_size, int) else tuple(kernel_size)
        self.in_channels, self.out_channels, self.kernel_size = in_channels, out_channels, k
        self.stride, self.padding, self.dilation = stride, padding, dilation
        self.groups, self.deform_groups = groups, deform_groups
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *k))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        n = in_channels * k[0] * k[1]
        stdv = 1.0 / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.zero_()
