"""ctypes wrapper around oracle/dcn_ref.c (TEST INFRASTRUCTURE)."""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libdcn_ref.so")


def build(force=False):
    src = os.path.join(_HERE, "dcn_ref.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "_build/libdcn_ref.so"])
    return _SO


def modulated_deform_conv2d_c(x, offset, mask, weight, bias, stride=1, padding=0, dilation=1,
                              groups=1, deform_groups=1):
    assert groups == 1
    lib = ctypes.CDLL(build())
    N, C, H, W = x.shape
    Co, _, kh, kw = weight.shape
    Ho = (H + 2 * padding - (dilation * (kh - 1) + 1)) // stride + 1
    Wo = (W + 2 * padding - (dilation * (kw - 1) + 1)) // stride + 1
    arrs = [np.ascontiguousarray(t.detach().float().numpy()) for t in (x, offset, mask, weight)]
    b = None if bias is None else np.ascontiguousarray(bias.detach().float().numpy())
    out = np.empty((N, Co, Ho, Wo), np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    p = lambda a: a.ctypes.data_as(fp) if a is not None else None
    rc = lib.dcn_ref_forward(p(arrs[0]), p(arrs[1]), p(arrs[2]), p(arrs[3]), p(b), p(out),
                             N, C, H, W, Co, kh, kw, stride, padding, dilation, deform_groups)
    assert rc == 0
    return torch.from_numpy(out)
