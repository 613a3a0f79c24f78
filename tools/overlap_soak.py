"""Soak version of tests/test_gpu_model.py::test_stream_overlap_is_bit_identical_to_serial: N forwards of the 432x240 T=10
clip with SPyNet on the side stream, every one compared bit for bit with the single-stream result.
    python tools/overlap_soak.py [N=300]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd.engine import Engine
from e2fgvi_amd.synth import synth_clip, synth_state_dict
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
bad = 0
for kind in ("stress", "default"):
    eng = Engine(synth_state_dict("e2fgvi", kind, 0), "e2fgvi", dev)
    x = synth_clip(1, 10, 240, 432, seed=3, moving=True)[0].to(dev)
    eng.overlap_flows = False
    base, (bf, bb) = eng.forward(x, 10)
    torch.cuda.synchronize()
    eng.overlap_flows = True
    for i in range(n // 2):
        got, (ff, fb) = eng.forward(x, 10)
        torch.cuda.synchronize()
        if not (torch.equal(ff, bf) and torch.equal(fb, bb) and torch.equal(got, base)):
            bad += 1
    print("%s weights: %d overlapped forwards, %d differ from the serial result" % (kind, n // 2, bad), flush=True)
sys.exit(1 if bad else 0)
