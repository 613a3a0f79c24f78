"""Time the HQ model at an arbitrary resolution (synthetic clip, random-init weights).
    python tools/hq_run.py 720x1296 10 [iters]"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, importlib
from e2fgvi_amd.synth import synth_clip, synth_state_dict
H, W = [int(v) for v in sys.argv[1].split("x")]
t = int(sys.argv[2]); iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
prec = sys.argv[4] if len(sys.argv) > 4 else "fp32"
dev = torch.device("cuda:0")
net = importlib.import_module("model.e2fgvi_hq").InpaintGenerator()
net.load_state_dict(synth_state_dict("e2fgvi_hq", "default", 0)); net = net.to(dev).eval()
net.precision = prec
x = synth_clip(1, t, H, W, seed=9)[0].to(dev)
out, _ = net(x, t); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    out, _ = net(x, t)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / iters
print(json.dumps({"hw": [H, W], "t": t, "precision": prec, "ms_per_forward": round(ms, 2), "frames_per_s": round(1e3 * t / ms, 2),
                  "finite": bool(torch.isfinite(out).all()), "max_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 2)}))
