"""Per-stage device time of one forward (hip events on the launch stream).  Usage:
    python tools/stage_times.py [--b 1] [--t 10] [--lt 10] [--iters 5]"""
import argparse, os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd.engine import Engine, token_grid
from e2fgvi_amd.synth import synth_clip, synth_state_dict

ap = argparse.ArgumentParser()
ap.add_argument("--b", type=int, default=1); ap.add_argument("--t", type=int, default=10)
ap.add_argument("--lt", type=int, default=10); ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--model", default="e2fgvi"); ap.add_argument("--hw", default="240x432")
a = ap.parse_args()
H, W = [int(v) for v in a.hw.split("x")]
dev = torch.device("cuda:0")
eng = Engine(synth_state_dict(a.model, "default", 0), a.model, dev)
x = synth_clip(a.b, a.t, H, W, seed=5)[0].to(dev)
b, t, lt = a.b, a.t, a.lt
h, w = H // 4, W // 4
fh, fw = token_grid(h, w)

def timed(fn, *args):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = fn(*args); e1.record(); torch.cuda.synchronize()
    return r, e0.elapsed_time(e1)

acc = {}
for it in range(a.iters + 1):
    rec = {}
    (fwd, bwd), rec["flows"] = timed(eng.flows, x, lt)
    enc, rec["encoder"] = timed(eng.encode, x)
    loc = enc.view(b, t, h, w, 128)[:, :lt].permute(1, 0, 2, 3, 4).contiguous()
    prop, rec["propagate"] = timed(eng.propagate, loc, fwd, bwd)
    tok, rec["soft_split"] = timed(lambda: eng.soft_split(enc).view(-1, 512))
    rec["blocks"] = 0.0
    for i in range(8):
        (tok, _), ms = timed(eng.block, i, tok, b, t, fh, fw, (h, w))
        rec["blocks"] += ms
    dec_in, rec["compose"] = timed(eng.compose, tok, enc, b, t, fh, fw)
    out, rec["decode"] = timed(eng.decode, dec_in)
    if it:
        for k, v in rec.items():
            acc[k] = acc.get(k, 0.0) + v / a.iters
tot = sum(acc.values())
print(json.dumps({"config": vars(a), "ms": {k: round(v, 3) for k, v in acc.items()}, "total_ms": round(tot, 3),
                  "frames_per_s_if_serial": round(1e3 * b * t / tot, 1)}))
