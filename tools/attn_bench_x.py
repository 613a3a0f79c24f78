"""Micro-benchmark of the bf16 fused focal attention at the e2fgvi_hq shapes (720x1296 T=10: 60x108 tokens; 1080x1944 T=20).
    python tools/attn_bench_x.py [fhxfw] [T]          (E2FGVI_LIB=<other build> for A/B runs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops
from e2fgvi_amd.engine import build_key_table
from e2fgvi_amd.synth import rolled_valid_index

dev = torch.device("cuda:0")
fh, fw = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "60x108").split("x"))
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
B = 1
rows, nwin = B * T * fh * fw, (fh // 5) * (fw // 9)
both = (torch.randn(rows + B * T * nwin, 1536, device=dev) * 0.5).bfloat16()
qkv, kvp = both[:rows], both[rows:]
tab, nk = build_key_table(fh, fw, rolled_valid_index().tolist())
tab, nk = torch.from_numpy(tab).to(dev), torch.from_numpy(nk).to(dev)
out = torch.empty(rows, 512, device=dev, dtype=torch.bfloat16)
gflop = B * 4 * (45 * T) * 128 * 2 * 2 * float(nk.float().sum().item()) * T * 1e-9
for rep in range(3):
    ops.focal_attention_bf16(qkv, kvp, tab, nk, B, T, fh, fw, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.focal_attention_bf16(qkv, kvp, tab, nk, B, T, fh, fw, out=out)
    e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 10
    print("attention bf16 %dx%d T=%d (%s): %8.1f us  %6.1f TF/s (valid-key flops)  checksum %.4f" % (
        fh, fw, T, os.environ.get("E2FGVI_LIB", "default lib"), us, gflop / us * 1e3, out.float().abs().mean().item()), flush=True)
