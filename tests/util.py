"""Shared helpers for the parity tests.

Determinism: every test draws its data from `gen(seed)` with a literal seed or `name_seed(case_name)` (zlib.crc32 -- NOT
Python's per-process salted `hash()`), so a test sees the same tensors in every process and on every box.  The env var
E2FGVI_TEST_SEED shifts all of them together: `tools/gpu_suite_soak.sh` runs the whole GPU suite under several shifts and
PYTHONHASHSEEDs to measure how far below its bound every comparison sits (E2FGVI_TEST_MARGINS=<file> makes `assert_close`
append `what, measured, allowed` lines)."""
import math
import os
import zlib

import torch

SEED_SHIFT = int(os.environ.get("E2FGVI_TEST_SEED", "0"))


def gen(seed):
    g = torch.Generator()
    g.manual_seed(int(seed) + 100003 * SEED_SHIFT)
    return g


def name_seed(name, salt=0):
    """a process-independent seed from a test-case name"""
    return zlib.crc32(name.encode()) % 100000 + salt


def fp32_tol(K, c=8.0, floor=2e-5):
    """allowed max|got - ref| / rms(ref) of an fp32 kernel that accumulates K products, against an fp64-computed reference:
    one fp32 rounding per accumulation step gives an rms error of ~ sqrt(K) * 2^-24 of the output rms; the maximum over
    1e5..1e6 outputs sits 4.5-5 sigma out; c = 8 leaves the rest as margin.  The floor covers the epilogue (bias, residual,
    activation, the Winograd transforms' own roundings)."""
    return max(floor, c * math.sqrt(K) * 2.0 ** -24)


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


def nchw(x):
    return x.permute(0, 3, 1, 2).contiguous()


def err(a, b):
    """(max abs error, error relative to the rms of the reference)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    d = (a - b).abs().max().item()
    rms = b.pow(2).mean().sqrt().item()
    return d, d / max(rms, 1e-12)


def _margin(what, measured, allowed):
    path = os.environ.get("E2FGVI_TEST_MARGINS")
    if path:
        with open(path, "a") as f:
            f.write("%s\t%.4e\t%.4e\t%.3f\n" % (what.replace("\t", " "), measured, allowed, measured / max(allowed, 1e-30)))


def assert_close(got, ref, rel, what=""):
    """max|got-ref| <= rel * rms(ref)  -- scale-free, so it bites at default init too."""
    d, r = err(got, ref)
    assert tuple(got.shape) == tuple(ref.shape), (what, tuple(got.shape), tuple(ref.shape))
    assert torch.isfinite(got.detach().float().cpu()).all(), what + ": non-finite output"
    _margin(what, r, rel)
    assert r <= rel, "%s: max abs err %.3e = %.3e x rms(ref) (allowed %.1e)" % (what, d, r, rel)
    return d, r


def assert_bound(value, bound, what):
    """value <= bound, recorded in the margin log like assert_close"""
    _margin(what, float(value), float(bound))
    assert value <= bound, "%s: %.3e exceeds the allowed %.3e" % (what, value, bound)


def assert_close_bf16(got, ref, what="", ulps=1.0, abs_rms=4e-3):
    """for results stored as bf16: |got - ref| <= ulps * 2^-8 * |ref| + abs_rms * rms(ref), element by element (one bf16
    rounding is 2^-9 relative; the default allows it twice) -- a max-abs bound relative to the rms alone would be set by the
    few largest elements."""
    got = got.detach().float().cpu()
    ref = ref.detach().float().cpu()
    assert tuple(got.shape) == tuple(ref.shape), (what, tuple(got.shape), tuple(ref.shape))
    assert torch.isfinite(got).all(), what + ": non-finite output"
    rms = ref.pow(2).mean().sqrt().item()
    allowed = (ulps * 2.0 ** -8) * ref.abs() + abs_rms * rms
    ratio = ((got - ref).abs() / allowed).max().item()
    _margin(what + " [bf16 elementwise]", ratio, 1.0)
    excess = ((got - ref).abs() - allowed).max().item()
    assert excess <= 0, "%s: error exceeds %.1f x 2^-8 relative + %.1e x rms by %.3e" % (what, ulps, abs_rms, excess)


import contextlib


@contextlib.contextmanager
def timed_tuning():
    """timing-based kernel selection (E2FGVI_AUTOTUNE=1) for the duration of a test: the default is the checked-in decision
    table (e2fgvi_amd/ops.py).  Decisions taken here are dropped on exit (the table and the nearest-size-class cache are restored:
    advisor finding of round 4 -- otherwise the kernels a later test of the same process runs would depend on the test order)"""
    from e2fgvi_amd import ops
    saved, tuned, nearest = ops.AUTOTUNE, dict(ops._TUNED), dict(ops._NEAREST)
    ops.AUTOTUNE = True
    try:
        yield
    finally:
        ops.AUTOTUNE = saved
        ops._TUNED.clear()
        ops._TUNED.update(tuned)
        ops._NEAREST.clear()
        ops._NEAREST.update(nearest)


def golden_case(path):
    """(npz, model, weights kind, clip, l_t, out stride, flow stride) of a fixture of tests/golden/make_golden.py: the clip is rebuilt
    from the seed in `meta` exactly as the generator built it (fixtures named *benchclip* hold bench.py's own clip: static box,
    unsmoothed noise)."""
    import os
    import numpy as np
    from e2fgvi_amd.synth import synth_clip
    z = np.load(path)
    H, W, t, lt, b, seed, so, sf = [int(v) for v in z["meta"]]
    if "benchclip" in os.path.basename(path):
        x, _ = synth_clip(b, t, H, W, seed=seed, smooth=False)
    else:
        x, _ = synth_clip(b, t, H, W, seed=seed, moving=True)
    return z, str(z["model"]), str(z["kind"]), x, lt, so, sf
