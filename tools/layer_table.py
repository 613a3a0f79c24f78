"""Per-launch table of one graph-free forward: layer, shape, kernel, device time, algorithmic and issued GFLOP.

    python tools/layer_table.py [--model e2fgvi] [--hw 240x432] [--t 10] [--precision fp32] [--out gpurun_out/layer_table]

Every C-ABI launch of the forward is bracketed by hip events on the launch stream (e2fgvi_amd.lib tracing); the MFMA
kernels carry the work accounting of ops.PackedConv._work / PackedDcn / focal_attention.  The tile choices are the ones
the benchmark uses (the tuning pass runs in the warm-up forwards, before tracing starts).  Three traced forwards, the
per-launch median is reported.  `frac` of bench.py can be recomputed from the tracked copy under profiles/:
sum(issued GFLOP) / (forward time) / peak."""
import argparse
import importlib
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from e2fgvi_amd import lib
from e2fgvi_amd.synth import synth_clip, synth_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="e2fgvi")
ap.add_argument("--hw", default="240x432")
ap.add_argument("--t", type=int, default=10)
ap.add_argument("--clips", type=int, default=1)
ap.add_argument("--precision", default="fp32")
ap.add_argument("--out", default="gpurun_out/layer_table")
a = ap.parse_args()
H, W = [int(v) for v in a.hw.split("x")]
dev = torch.device("cuda:0")
net = importlib.import_module("model." + a.model).InpaintGenerator()
net.load_state_dict(synth_state_dict(a.model, "default", 0))
net = net.to(dev).eval()
net.precision = a.precision
x = synth_clip(a.clips, a.t, H, W, seed=0, smooth=False)[0].to(dev)
for _ in range(3):
    net(x, a.t)
torch.cuda.synchronize()
serial = net.engine().overlap_flows
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    net(x, a.t)
e1.record()
torch.cuda.synchronize()
fwd_ms = e0.elapsed_time(e1) / 5
runs = []
net.engine().overlap_flows = False      # traced launches are timed one by one: no second stream sharing the CUs
for _ in range(3):
    lib.TRACE = []
    net(x, a.t)
    torch.cuda.synchronize()
    runs.append(lib.TRACE)
    lib.TRACE = None
net.engine().overlap_flows = serial
assert len({len(r) for r in runs}) == 1
rows = []
for k, r in enumerate(runs[0]):
    us = statistics.median(1e3 * q[k]["e0"].elapsed_time(q[k]["e1"]) for q in runs)
    m = r["meta"] or {}
    row = {"i": k, "symbol": r["symbol"].replace("e2fgvi_", ""), "layer": m.get("layer", ""), "kernel": m.get("kernel", ""),
           "shape": m.get("shape", ""), "us": round(us, 2)}
    if "macs" in m:
        row["gflop"] = round(2e-9 * m["macs"], 3)
        row["gflop_issued"] = round(2e-9 * m["issued"], 3)
        row["tflops"] = round(2e-9 * m["macs"] / us * 1e3, 1) if us > 0 else None
        row["tflops_issued"] = round(2e-9 * m["issued"] / us * 1e3, 1) if us > 0 else None
    rows.append(row)
mf = [r for r in rows if "gflop" in r]
summary = {"model": a.model, "hw": [H, W], "t": a.t, "clips": a.clips, "precision": a.precision,
           "forward_ms_untraced": round(fwd_ms, 3), "launches": len(rows), "mfma_launches": len(mf),
           "sum_us_all": round(sum(r["us"] for r in rows), 1), "sum_us_mfma": round(sum(r["us"] for r in mf), 1),
           "gflop_algorithmic": round(sum(r["gflop"] for r in mf), 1), "gflop_issued": round(sum(r["gflop_issued"] for r in mf), 1)}
peak = 157.3 if a.precision == "fp32" else 2500.0
summary["tflops_issued_whole_forward"] = round(summary["gflop_issued"] / fwd_ms, 1)
summary["frac_issued"] = round(summary["gflop_issued"] / fwd_ms / peak, 4)
eff_peak = peak if (a.precision != "fp32" or os.environ.get("E2FGVI_X3", "1") == "0") else 2500.0 / 6
summary["effective_peak"] = round(eff_peak, 1)       # the rate the pipe retires fp32 products in this arithmetic (bench.py EFFECTIVE_PEAK)
summary["frac_effective"] = round(summary["gflop_algorithmic"] / fwd_ms / eff_peak, 4)
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
json.dump({"summary": summary, "rows": rows}, open(a.out + ".json", "w"), indent=0)
# grouped view: same (layer-name-without-index, kernel, shape) collapsed
groups = {}
for r in rows:
    import re
    key = (re.sub(r"transformer\.\d+", "transformer.*", r["layer"] or r["symbol"]), r["kernel"], r["shape"])
    g = groups.setdefault(key, {"n": 0, "us": 0.0, "gflop": 0.0, "iss": 0.0})
    g["n"] += 1; g["us"] += r["us"]; g["gflop"] += r.get("gflop", 0.0); g["iss"] += r.get("gflop_issued", 0.0)
with open(a.out + ".md", "w") as f:
    f.write("# per-layer table, one graph-free forward (%s)\n\n```\n%s\n```\n\n" % (" ".join(sys.argv[1:]), json.dumps(summary, indent=1)))
    f.write("| layer | kernel | shape | launches | total us | us/launch | GFLOP alg | GFLOP issued | TF/s alg | TF/s issued |\n|---|---|---|---|---|---|---|---|---|---|\n")
    for (layer, kern, shape), g in sorted(groups.items(), key=lambda kv: -kv[1]["us"]):
        f.write("| %s | %s | %s | %d | %.1f | %.1f | %.2f | %.2f | %s | %s |\n" % (
            layer, kern, shape, g["n"], g["us"], g["us"] / g["n"], g["gflop"], g["iss"],
            ("%.1f" % (g["gflop"] / g["us"] * 1e3)) if g["gflop"] else "", ("%.1f" % (g["iss"] / g["us"] * 1e3)) if g["iss"] else ""))
print(json.dumps(summary))
