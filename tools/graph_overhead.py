"""Per-node cost of HIP-graph replay on this box: N dependent launches of (a) an empty-ish kernel, (b) a ~10 us kernel, captured
in one graph and replayed; eager launches of the same for comparison.    python tools/graph_overhead.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(n, numel, reps=20):
    dev = torch.device("cuda:0")
    x = torch.zeros(numel, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            for _ in range(n):
                x.add_(1.0)
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            for _ in range(n):
                x.add_(1.0)
        s.synchronize()
        eager = (time.perf_counter() - t0) / reps
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                x.add_(1.0)
        for _ in range(3):
            g.replay()
        s.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        s.synchronize()
        graph = (time.perf_counter() - t0) / reps
        # one launch alone, device time
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(200):
            x.add_(1.0)
        e1.record(s)
        e1.synchronize()
    return eager * 1e6 / n, graph * 1e6 / n, e0.elapsed_time(e1) * 1e3 / 200


for numel in (64, 1 << 20, 1 << 24, 1 << 26):
    e, g, k = run(300, numel)
    print("x.add_(1) on %9d floats, 300 dependent launches: eager %.2f us/launch, graph replay %.2f us/launch, back-to-back eager stream %.2f us/launch"
          % (numel, e, g, k), flush=True)
