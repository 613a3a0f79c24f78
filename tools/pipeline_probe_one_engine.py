import sys, time, importlib, torch
sys.path.insert(0, "/root/repo")
from e2fgvi_amd import runner
from e2fgvi_amd.synth import synth_clip, synth_state_dict
dev = torch.device("cuda:0")
net = importlib.import_module("model.e2fgvi").InpaintGenerator()
net.load_state_dict(synth_state_dict("e2fgvi", "default", 0))
net = net.to(dev).eval()
x = synth_clip(1, 10, 240, 432, seed=0, smooth=False)[0].to(dev)
for k in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,3,2").split(",")]:
    st = runner.ShardedStep(net, x, 10, in_flight=k)
    for _ in range(3 + k):
        st.run()
    st.finish(); torch.cuda.synchronize()
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(40):
            st.run()
        st.finish(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("one engine, in flight %d: %.3f ms/step %.1f frames/s" % (k, dt * 1e3 / 40, 400 / dt), flush=True)
