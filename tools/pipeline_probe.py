"""Two (or more) forwards in flight: K independent HIP graphs of the one-clip forward replayed round-robin on K streams against the
same graph replayed back to back on one stream.  Each pipeline has its own engine (own static buffers), the weights are the same.

    gpurun -- 'python tools/pipeline_probe.py --k 1,2,3 > gpurun_out/pipeline_probe.txt'
"""
import argparse
import importlib
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--k", default="1,2,3")
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--model", default="e2fgvi")
    ap.add_argument("--hw", default="240x432")
    ap.add_argument("--t", type=int, default=10)
    ap.add_argument("--lt", type=int, default=10)
    ap.add_argument("--clips", type=int, default=1)
    ap.add_argument("--precision", default="fp32")
    a = ap.parse_args()
    from e2fgvi_amd import runner
    from e2fgvi_amd.synth import synth_clip, synth_state_dict
    dev = torch.device("cuda:0")
    H, W = [int(v) for v in a.hw.split("x")]
    kmax = max(int(v) for v in a.k.split(","))
    sd = synth_state_dict(a.model, "stress", 0)
    nets, steps, streams, xs = [], [], [], []
    for i in range(kmax):
        net = importlib.import_module("model." + a.model).InpaintGenerator()
        net.load_state_dict(sd)
        net = net.to(dev).eval()
        net.precision = a.precision
        x = synth_clip(a.clips, a.t, H, W, seed=i, smooth=False)[0].to(dev)
        st = runner.ShardedStep(net, x, a.lt, group_world=1)
        for _ in range(3):
            st.run()
        torch.cuda.synchronize()
        nets.append(net); steps.append(st); streams.append(torch.cuda.Stream()); xs.append(x)
    refs = [st.run().clone() for st in steps]
    torch.cuda.synchronize()
    for k in [int(v) for v in a.k.split(",")]:
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(a.steps):
                j = i % k
                with torch.cuda.stream(streams[j]):
                    steps[j].run()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            ok = all(torch.equal(steps[j].out, refs[j]) for j in range(k))
            print("in flight %d: %d steps in %.1f ms = %.3f ms per step, %.1f frames/s  (outputs bit-equal to the serial run: %s)"
                  % (k, a.steps, dt * 1e3, dt * 1e3 / a.steps, a.clips * a.t * a.steps / dt, ok), flush=True)


if __name__ == "__main__":
    main()
