"""mmcv.cnn stand-in (test infrastructure).  Used at flow_comp.py:181-215, feat_prop.py:33."""
import torch.nn as nn


def constant_init(module, val, bias=0):
    if getattr(module, "weight", None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, "bias", None) is not None:
        nn.init.constant_(module.bias, bias)


class ConvModule(nn.Module):
    """conv (+ReLU) with mmcv's child names ``conv`` / ``activate`` and mmcv's default
    init (kaiming normal, fan_out, relu gain; zero bias)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 norm_cfg=None, act_cfg=dict(type="ReLU")):
        super().__init__()
        assert norm_cfg is None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding)
        self.with_activation = act_cfg is not None
        if self.with_activation:
            assert act_cfg["type"] == "ReLU"
            self.activate = nn.ReLU(inplace=True)
        nn.init.kaiming_normal_(self.conv.weight, a=0, mode="fan_out", nonlinearity="relu")
        nn.init.constant_(self.conv.bias, 0)

    def forward(self, x):
        x = self.conv(x)
        if self.with_activation:
            x = self.activate(x)
        return x
