"""The one-clip forward (HIP graph, SPyNet + propagation split on the side stream) against HIP's stream -> hardware-queue mapping:
GPU_MAX_HW_QUEUES (environment) and the number of pool streams taken before the engine takes its side stream (--burn)."""
import sys, time, importlib, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from e2fgvi_amd import runner
from e2fgvi_amd.synth import synth_clip, synth_state_dict
burn = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device("cuda:0")
keep = [torch.cuda.Stream() for _ in range(burn)]
net = importlib.import_module("model.e2fgvi").InpaintGenerator()
net.load_state_dict(synth_state_dict("e2fgvi", "default", 0))
net = net.to(dev).eval()
x = synth_clip(1, 10, 240, 432, seed=0, smooth=False)[0].to(dev)
st = runner.ShardedStep(net, x, 10)
for _ in range(4):
    st.run()
torch.cuda.synchronize()
best = 1e9
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(30):
        st.run()
    torch.cuda.synchronize()
    best = min(best, (time.perf_counter() - t0) / 30)
print("GPU_MAX_HW_QUEUES=%s burn=%d: %.3f ms/step %.1f frames/s" % (os.environ.get("GPU_MAX_HW_QUEUES", "default"), burn, best * 1e3, 10 / best), flush=True)
