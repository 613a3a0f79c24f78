"""Fabric-side traffic of one forward BY STAGE: the forward runs stage by stage on one stream (SPyNet,
encoder, propagation, soft split, the eight blocks, soft composite, decoder -- the engine's own stage methods, the same kernels
the table selects for the whole forward), a marker kernel (torch.arange: no other launch of the process has that name) between
stages, under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, kernel trace only).  --parse sums the
counters between the markers of the LAST forward of the process.

    bash tools/traffic_by_stage.sh r06            -> gpurun_out/traffic_r06/ + profiles/r06_traffic_by_stage.md
"""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

STAGES = ["spynet", "encoder", "propagation", "soft split"] + ["block %d" % i for i in range(8)] + ["soft composite", "decoder"]


def workload(a):
    import importlib
    import torch
    from e2fgvi_amd.engine import token_grid
    from e2fgvi_amd.synth import synth_clip, synth_state_dict
    dev = torch.device("cuda:0")
    H, W = [int(v) for v in a.hw.split("x")]
    net = importlib.import_module("model." + a.model).InpaintGenerator()
    net.load_state_dict(synth_state_dict(a.model, "default", 0))
    net = net.to(dev).eval()
    net.precision = a.precision
    x = synth_clip(1, a.t, H, W, seed=0, smooth=False)[0].to(dev)
    lt = a.lt or a.t
    eng = net.engine()
    for _ in range(2):
        net(x, lt)                                   # weight packing, tables
    torch.cuda.synchronize()
    b, t = 1, a.t
    h, w = H // 4, W // 4
    fh, fw = token_grid(h, w)
    bf16 = a.precision == "bf16"

    def mark():
        torch.arange(17, device=dev)

    for rep in range(2):                             # the second pass is the one that is parsed
        mark()
        fwd, bwd = eng.flows(x, lt)
        mark()
        enc = eng.encode_x(x) if bf16 else eng.encode(x)
        mark()
        ch = enc.shape[3]
        enc5 = enc.view(b, t, h, w, ch)
        loc = enc5[0, :lt].unsqueeze(1)
        if bf16:
            prop = eng.propagate_x(loc, fwd, bwd)
            enc5[0, :lt].copy_(prop[:, 0])
        else:
            eng.propagate(loc, fwd, bwd, inplace=True)
        mark()
        tok = (eng.xss([enc], out_dtype=torch.float32) if bf16 else eng.soft_split(enc)).view(b * t * fh * fw, 512)
        mark()
        tok16 = None
        for i in range(8):
            if bf16:
                tok, _, tok16 = eng.block_x(i, tok, b, t, fh, fw, (h, w), want_bf16_copy=(i == 7))
            else:
                tok, _ = eng.block(i, tok, b, t, fh, fw, (h, w))
            mark()
        dec_in = eng.compose_x(tok16, enc, b, t, fh, fw) if bf16 else eng.compose(tok, enc, b, t, fh, fw)
        mark()
        out = eng.decode_x(dec_in) if bf16 else eng.decode(dec_in)
        mark()
        torch.cuda.synchronize()
    print("frames", tuple(out.shape), float(out.abs().mean()))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0][:64]


def parse(a):
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        rows = []
        for f in glob.glob(os.path.join(a.parse, c, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r.get("Counter_Name") == c:
                    rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
        rows.sort()
        marks = [k for k, r in enumerate(rows) if "arange" in r[1]]
        assert len(marks) >= len(STAGES) + 1, "markers not found: %d" % len(marks)
        marks = marks[-(len(STAGES) + 1):]
        per = []
        for s, (lo, hi) in zip(STAGES, zip(marks, marks[1:])):
            seg = rows[lo + 1:hi]
            kib = sum(v for _, _, v in seg)
            byk = collections.Counter()
            for _, n, v in seg:
                byk[short(n)] += v
            per.append((s, len(seg), kib, byk))
        res[c] = per
    lines, js = [], {"stages": []}
    lines.append("| Stage | launches | read GB (FETCH_SIZE x 2) | written GB (WRITE_SIZE) | total GB | share | largest contributors (GB) |")
    lines.append("|---|---|---|---|---|---|---|")
    tot = 0.0
    stage_tot = []
    for (s, n, fk, fby), (_, _, wk, wby) in zip(res["FETCH_SIZE"], res["WRITE_SIZE"]):
        stage_tot.append((fk * 2048 + wk * 1024) / 1e9)
    tot = sum(stage_tot)
    blocks = [k for k, s in enumerate(STAGES) if s.startswith("block")]
    for k, ((s, n, fk, fby), (_, _, wk, wby)) in enumerate(zip(res["FETCH_SIZE"], res["WRITE_SIZE"])):
        both = collections.Counter()
        for n_, v in fby.items():
            both[n_] += v * 2048 / 1e9
        for n_, v in wby.items():
            both[n_] += v * 1024 / 1e9
        top = ", ".join("%s %.2f" % (n_, v) for n_, v in both.most_common(3))
        js["stages"].append({"stage": s, "launches": n, "read_gb": fk * 2048 / 1e9, "written_gb": wk * 1024 / 1e9, "by_kernel_gb": dict(both)})
        lines.append("| %s | %d | %.3f | %.3f | %.3f | %.1f %% | %s |" % (s, n, fk * 2048 / 1e9, wk * 1024 / 1e9, stage_tot[k], 100 * stage_tot[k] / tot, top))
    lines.append("| **forward** | %d | %.3f | %.3f | **%.3f** | 100 %% | blocks together %.3f |" % (
        sum(r[1] for r in res["FETCH_SIZE"]), sum(r[2] for r in res["FETCH_SIZE"]) * 2048 / 1e9,
        sum(r[2] for r in res["WRITE_SIZE"]) * 1024 / 1e9, tot, sum(stage_tot[k] for k in blocks)))
    js["total_gb"] = tot
    try:
        from e2fgvi_amd import lib
        js["library_sha16"] = lib.library_key()
    except Exception:
        pass
    text = "\n".join(lines)
    print(text)
    if a.out:
        with open(a.out + ".md", "w") as fh_:
            fh_.write("# Fabric-side traffic of one forward by stage (%s)\n\n%s\n\n%s\n" % (a.title, a.note, text))
        json.dump(js, open(a.out + ".json", "w"), indent=1)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="e2fgvi")
    ap.add_argument("--hw", default="240x432")
    ap.add_argument("--t", type=int, default=10)
    ap.add_argument("--lt", type=int, default=0)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--parse", default="")
    ap.add_argument("--out", default="")
    ap.add_argument("--title", default="")
    ap.add_argument("--note", default="rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KiB; the read side doubled: gfx950 "
                                       "calibration of MI355X_MICROARCH.md), summed over the dispatches between marker kernels of a forward "
                                       "run stage by stage on one stream (tools/traffic_by_stage.py).  Fabric-side: Infinity-Cache hits are "
                                       "counted, so this is traffic between the L2s and the memory side, not DRAM bytes.")
    a = ap.parse_args()
    if a.parse:
        parse(a)
    else:
        workload(a)
