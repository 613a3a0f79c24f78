"""decoder.6 (64 -> 3, 3x3) on the tail kernel vs the general kernels, device time per launch.
    python tools/tail_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from e2fgvi_amd import ops


def timeit(fn, reps=20):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    w = (torch.randn(3, 64, 3, 3, generator=g) / 24).to(dev)
    b = torch.zeros(3, device=dev)
    for (N, H, W) in ((10, 240, 432), (10, 720, 1296), (20, 1080, 1944)):
        for dt in (torch.float32, torch.bfloat16):
            x = torch.randn(N, H, W, 64, device=dev).to(dt)
            out = torch.empty((N, 3, H, W), device=dev)
            tail = ops.PackedTailConv(w, b, dtype=dt)
            gen = ops.PackedConv(w, b, [64], pad=1) if dt == torch.float32 else ops.PackedConvX(w, b, [64], pad=1)
            t_tail = timeit(lambda: tail([x], act=ops.ACT_TANH, out=out))
            t_gen = timeit(lambda: gen([x], act=ops.ACT_TANH, out_nchw=True, out=out))
            byts = x.numel() * x.element_size() + out.numel() * 4
            print("N%d %dx%d %s: tail %.1f us (%.2f TB/s algorithmic)   general %.1f us" % (
                N, H, W, "fp32" if dt == torch.float32 else "bf16", t_tail, byts / t_tail / 1e6, t_gen), flush=True)
            del x, out
            if N * H * W > 3e7 and dt == torch.float32:
                pass


if __name__ == "__main__":
    main()
