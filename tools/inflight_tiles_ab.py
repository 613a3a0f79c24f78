"""Forwards in flight with the one-frame propagation layers on WIDER workgroups (fewer CUs per launch, longer launch):
python tools/inflight_tiles_ab.py   (one box, one process per row)
rows: K, block-shape code of the split-operand Winograd kernel forced on the 128-cout 3x3 layers of one 60x108 frame
(0 = the table's 132: 8x16 pixels x 32 couts = 224 workgroups; 164: x 64 couts = 112; 6064: 16x16 x 64 = 56)"""
import sys, time, importlib, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import torch
    from e2fgvi_amd import runner, ops
    from e2fgvi_amd.synth import synth_clip, synth_state_dict
    k, code = int(sys.argv[1]), int(sys.argv[2])
    which = sys.argv[3] if len(sys.argv) > 3 else "all"
    dev = torch.device("cuda:0")
    net = importlib.import_module("model.e2fgvi").InpaintGenerator()
    net.load_state_dict(synth_state_dict("e2fgvi", "default", 0))
    net = net.to(dev).eval()
    x = synth_clip(1, 10, 240, 432, seed=0, smooth=False)[0].to(dev)
    ref = net(x, 10)[0].clone()
    if code:
        plain = ops._decision
        hits = {}

        def forced(key):
            if (key[-1] == "x3" and key[0] == 128 and key[2] == 3 and key[8] == 50
                    and (which == "all" or (which == "co0" and len(key[1]) == 4) or (which in ("res", "nores") and (which == "res") == bool(key[9])))):
                hits[key] = hits.get(key, 0) + 1
                return ops.W3_BASE + code
            return plain(key)
        ops._decision = forced
    st = runner.ShardedStep(net, x, 10, in_flight=k)
    for _ in range(4 + k):
        st.run()
    out = st.finish(); torch.cuda.synchronize()
    err = float((out - ref).abs().max())
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(30):
            st.run()
        st.finish(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 30)
    print("in flight %d  code %d (%s): %.3f ms/step %.1f frames/s  window %s  max |diff| to the table's forward %.2e  forced keys %d"
          % (k, code, which, best * 1e3, 10 / best, getattr(st, "stream_window", None), err, len(hits) if code else 0), flush=True)
else:
    rows = ((1, 0), (2, 0), (2, 164), (2, 6064), (3, 164), (3, 6064), (1, 164), (1, 6064), (2, 0), (2, 164, "res"), (2, 164, "nores"),
            (4, 6064))
    if os.environ.get("AB_ROWS") == "k3":          # second pass: is K = 3 with 112-workgroup layers ahead of K = 2 / K = 3 as tabled?
        rows = ((2, 0), (3, 0), (3, 164), (2, 0), (3, 0), (3, 164), (3, 164, "nores"), (3, 164, "res"), (4, 164))
    if os.environ.get("AB_ROWS") == "co0":         # the whole conv_offset.0 (388 -> 128) has no entry at one frame: it takes 164 from two clips
        rows = ((2, 0), (2, 132, "co0"), (2, 0), (2, 132, "co0"), (2, 0), (2, 132, "co0"), (3, 132, "co0"), (1, 0))
    for row in rows:
        subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(v) for v in row])
