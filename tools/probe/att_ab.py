"""Alternating A/B of two libraries on the fp32 split-operand attention at the headline shape (20x36 tokens, T=10), one process per library is
impossible (one library per process), so:  python tools/probe/att_ab.py <libA> <libB>   runs each library in a subprocess 3 times alternately
and prints the medians of 5 x 20 launches + a checksum of the output (bit-identity across libraries)."""
import os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import torch, hashlib
    from e2fgvi_amd import ops
    from e2fgvi_amd.engine import build_key_table
    from e2fgvi_amd.synth import rolled_valid_index
    dev = torch.device("cuda:0")
    fh, fw, T, B = 20, 36, 10, 1
    rows, nwin = B * T * fh * fw, (fh // 5) * (fw // 9)
    torch.manual_seed(0)
    both = torch.randn(rows + B * T * nwin, 1536, device=dev) * 0.5
    tab, nk = build_key_table(fh, fw, rolled_valid_index().tolist())
    tab, nk = torch.from_numpy(tab).to(dev), torch.from_numpy(nk).to(dev)
    planes = torch.empty(3, both.shape[0], 1024, dtype=torch.bfloat16, device=dev)
    ops.split3_kv(both, out=planes)
    out = torch.empty(rows, 512, device=dev)
    for _ in range(3):
        ops.focal_attention_x3(both[:rows], planes, tab, nk, B, T, fh, fw, out=out, waves=14)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.focal_attention_x3(both[:rows], planes, tab, nk, B, T, fh, fw, out=out, waves=14)
        e1.record(); torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1) / 20)
    print("RESULT %.2f %.2f %s" % (statistics.median(ts), min(ts), hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]))
else:
    libs = sys.argv[1:3]
    res = {l: [] for l in libs}
    sums = {}
    for rnd in range(3):
        for l in libs:
            env = dict(os.environ)
            if l != "product":
                env["E2FGVI_LIB"] = l
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, capture_output=True, text=True)
            line = [x for x in p.stdout.splitlines() if x.startswith("RESULT")]
            if not line:
                print(l, "failed:", p.stderr[-300:]); continue
            med, mn, h = line[0].split()[1:]
            res[l].append(float(med)); sums[l] = h
    for l in libs:
        print("%-50s median of medians %7.2f us  (%s)  checksum %s" % (l, statistics.median(res[l]), " ".join("%.1f" % v for v in res[l]), sums.get(l)))
