"""Where the bf16 data path's error comes from: every traced tensor of a bf16 forward (flows, propagated features, the token stream
behind each transformer block, the frames) against the CPU oracle's trace of the same clip, as rms of the difference / rms of the
reference, for the stress and the peaked weights.  Variants: --prop-fp32-state keeps the recurrent propagation state in fp32
(E2FGVI_PROP_BF16SRC=0).

    gpurun -- 'python tools/bf16_error_growth.py --hw 240x432 --t 6 --lt 4 > gpurun_out/bf16_error_growth.txt'
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="e2fgvi_hq")
    ap.add_argument("--hw", default="240x432")
    ap.add_argument("--t", type=int, default=6)
    ap.add_argument("--lt", type=int, default=4)
    ap.add_argument("--kinds", default="stress,peaked")
    ap.add_argument("--precision", default="bf16")
    a = ap.parse_args()
    import importlib
    from e2fgvi_amd.synth import synth_clip, synth_state_dict
    from oracle import e2fgvi_oracle as O
    H, W = [int(v) for v in a.hw.split("x")]
    dev = torch.device("cuda:0")
    torch.set_num_threads(16)
    for kind in a.kinds.split(","):
        sd = synth_state_dict(a.model, kind, 0)
        x, _ = synth_clip(1, a.t, H, W, seed=22, moving=True)
        tr = {}
        ref, (rf, rb) = O.forward(sd, x, a.lt, a.model, tr)
        net = importlib.import_module("model." + a.model).InpaintGenerator()
        net.load_state_dict(sd)
        net = net.to(dev).eval()
        net.precision = a.precision
        got = {}
        out, (ff, fb) = net.engine().forward(x.to(dev), a.lt, trace=got)
        torch.cuda.synchronize()
        b, t = 1, a.t
        h, w = H // 4, W // 4

        def rel(g, r):
            g, r = g.double().cpu(), r.double()
            return ((g - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt()).item(), (g - r).abs().max().item()
        rows = [("flow_fwd", ff, rf), ("flow_bwd", fb, rb)]
        if "prop" in got:
            rows.append(("prop (all frames)", got["prop"].view(b * t, h, w, -1).permute(0, 3, 1, 2), tr["prop"].reshape(b * t, -1, h, w)))
        for i in range(1, 9):
            if "tokens%d" % i in got:
                rows.append(("tokens%d" % i, got["tokens%d" % i].reshape(-1, 512), tr["tokens%d" % i].reshape(-1, 512)))
        rows.append(("frames", out, ref))
        print("== %s %s %dx%d T=%d l_t=%d, %s, PROP_BF16SRC=%s" % (a.model, kind, W, H, a.t, a.lt, a.precision, os.environ.get("E2FGVI_PROP_BF16SRC", "1")))
        for name, g, r in rows:
            rr, mx = rel(g, r)
            print("   %-20s rms(diff)/rms(ref) %.3e   max abs %.3e" % (name, rr, mx))
        del net


if __name__ == "__main__":
    main()
