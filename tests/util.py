"""Shared helpers for the parity tests."""
import torch


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def err(a, b):
    """(max abs error, error relative to the rms of the reference)."""
    a = a.detach().float().cpu()
    b = b.detach().float().cpu()
    d = (a - b).abs().max().item()
    rms = b.pow(2).mean().sqrt().item()
    return d, d / max(rms, 1e-12)


def assert_close(got, ref, rel, what=""):
    """max|got-ref| <= rel * rms(ref)  -- scale-free, so it bites at default init too."""
    d, r = err(got, ref)
    assert tuple(got.shape) == tuple(ref.shape), (what, tuple(got.shape), tuple(ref.shape))
    assert torch.isfinite(got.detach().float().cpu()).all(), what + ": non-finite output"
    assert r <= rel, "%s: max abs err %.3e = %.3e x rms(ref) (allowed %.1e)" % (what, d, r, rel)
    return d, r
