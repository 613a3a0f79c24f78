"""The CPU baseline of bench.py (oracle/e2fgvi_oracle.py, torch CPU fp32) at several intra-op thread counts on THIS host:
backs bench.py::cpu_baseline's choice of 16 threads (SURVEY.md 8d asks for "N = host cores"; on the GPU box's 256 hardware
threads torch's intra-op pool does not scale on these small ops).   python tools/cpu_threads.py [threads ...]"""
import os, sys, time, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd.synth import synth_clip, synth_state_dict
from oracle import e2fgvi_oracle as O

host = os.cpu_count() or 1
counts = [int(v) for v in sys.argv[1:]] or [c for c in (8, 16, 32, 64, 128, 256) if c <= host]
sd = synth_state_dict("e2fgvi", "default", 0)
x, _ = synth_clip(1, 10, 240, 432, seed=0, smooth=False)
print("host threads: %d; workload: e2fgvi 432x240 T=10 l_t=10, one clip; 1 warm-up + median of 3 (2 above 30 s)" % host, flush=True)
for n in counts:
    torch.set_num_threads(n)
    ts = []
    for k in range(4):
        t0 = time.perf_counter()
        O.forward(sd, x, 10, "e2fgvi")
        ts.append(time.perf_counter() - t0)
        if k >= 2 and ts[-1] > 30:
            break
    dt = statistics.median(ts[1:])
    print("threads %3d: %7.2f s per forward  %6.3f frames/s   (runs: %s)" % (n, dt, 10 / dt, ", ".join("%.1f" % t for t in ts)), flush=True)
