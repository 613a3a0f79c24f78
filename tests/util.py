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


def assert_close_bf16(got, ref, what="", ulps=1.0, abs_rms=4e-3):
    """for results stored as bf16: |got - ref| <= ulps * 2^-8 * |ref| + abs_rms * rms(ref), element by element (one bf16
    rounding is 2^-9 relative; the default allows it twice) -- a max-abs bound relative to the rms alone would be set by the
    few largest elements."""
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert tuple(got.shape) == tuple(ref.shape), (what, tuple(got.shape), tuple(ref.shape))
    assert torch.isfinite(got).all(), what + ": non-finite output"
    rms = ref.pow(2).mean().sqrt().item()
    excess = ((got - ref).abs() - (ulps * 2.0 ** -8) * ref.abs() - abs_rms * rms).max().item()
    assert excess <= 0, "%s: error exceeds %.1f x 2^-8 relative + %.1e x rms by %.3e" % (what, ulps, abs_rms, excess)
