"""fp32 focal attention: the fp32-MFMA kernel (attention.hip) against the split-operand kernel on the bf16 matrix pipe
(attention_x3.hip, + the split pass over the k / v columns) at the e2fgvi 432x240 T=10 shape (20x36 tokens) and the HQ grids.
    python tools/attn_bench_x3.py [fhxfw] [T]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops
from e2fgvi_amd.engine import build_key_table
from e2fgvi_amd.synth import rolled_valid_index

dev = torch.device("cuda:0")
fh, fw = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "20x36").split("x"))
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = 1
rows, nwin = B * T * fh * fw, (fh // 5) * (fw // 9)
torch.manual_seed(0)
both = torch.randn(rows + B * T * nwin, 1536, device=dev) * 0.5
qkv, kvp = both[:rows], both[rows:]
tab, nk = build_key_table(fh, fw, rolled_valid_index().tolist())
tab, nk = torch.from_numpy(tab).to(dev), torch.from_numpy(nk).to(dev)
gflop = B * 4 * (45 * T) * 128 * 2 * 2 * float(nk.float().sum().item()) * T * 1e-9


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


out = torch.empty(rows, 512, device=dev)
ref = ops.focal_attention(qkv, kvp, tab, nk, B, T, fh, fw).clone()
us = timed(lambda: ops.focal_attention(qkv, kvp, tab, nk, B, T, fh, fw, out=out))
print("attention fp32 %dx%d T=%d: fp32 MFMA kernel %8.1f us  %6.1f TF/s (valid-key flops)" % (fh, fw, T, us, gflop / us * 1e3), flush=True)
planes = torch.empty(3, both.shape[0], 1024, dtype=torch.bfloat16, device=dev)
us_s = timed(lambda: ops.split3_kv(both, out=planes))
print("  split3_kv of %d rows: %6.1f us (%.0f MB read + written)" % (both.shape[0], us_s, both.shape[0] * 1024 * 10e-6))
for w in (0, 2, 4, 8, 14):
    o = ops.focal_attention_x3(qkv, planes, tab, nk, B, T, fh, fw, waves=w)
    d = (o - ref).abs().max().item() / ref.pow(2).mean().sqrt().item()
    us = timed(lambda: ops.focal_attention_x3(qkv, planes, tab, nk, B, T, fh, fw, out=out, waves=w))
    print("  x3 kernel, %d waves: %8.1f us  %6.1f TF/s fp32-equivalent  (+ split: %7.1f us)  max |x3 - fp32| / rms %.2e"
          % (w, us, gflop / us * 1e3, us + us_s, d), flush=True)
