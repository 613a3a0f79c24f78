"""Forwards in flight against the engine's two-stream settings: python tools/inflight_ab.py  (one box, one process per row)
rows: K, prop split on/off, SPyNet on the side stream on/off"""
import sys, time, importlib, os, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    import torch
    from e2fgvi_amd import runner
    from e2fgvi_amd.synth import synth_clip, synth_state_dict
    k, split, overlap = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    clips = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    lt = int(sys.argv[5]) if len(sys.argv) > 5 else 10
    dev = torch.device("cuda:0")
    net = importlib.import_module("model.e2fgvi").InpaintGenerator()
    net.load_state_dict(synth_state_dict("e2fgvi", "default", 0))
    net = net.to(dev).eval()
    x = synth_clip(clips, 10, 240, 432, seed=0, smooth=False)[0].to(dev)
    net(x, lt)
    eng = net.engine()
    if not split:
        eng.prop_split = {}
    eng.overlap_flows = bool(overlap)
    st = runner.ShardedStep(net, x, lt, in_flight=k)
    for _ in range(4 + k):
        st.run()
    st.finish(); torch.cuda.synchronize()
    best = 1e9
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(30):
            st.run()
        st.finish(); torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 30)
    print("in flight %d  prop split %d  SPyNet on the side stream %d  clips %d l_t %d: %.3f ms/step %.1f frames/s  window %s"
          % (k, split, overlap, clips, lt, best * 1e3, clips * 10 / best, getattr(st, "stream_window", None)), flush=True)
else:
    for row in ((1, 1, 1), (1, 0, 1), (2, 1, 1), (2, 0, 1), (2, 0, 0), (2, 1, 0), (3, 0, 1), (3, 0, 0), (1, 0, 0)):
        subprocess.run([sys.executable, os.path.abspath(__file__)] + [str(v) for v in row])
