"""Micro-benchmark of the fused focal attention at the north-star shape (B=1,T=10,20x36 tokens)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops
from e2fgvi_amd.engine import build_key_table
from e2fgvi_amd.synth import rolled_valid_index
dev = torch.device("cuda:0")
B, T, fh, fw = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 10, 20, 36
rows, nwin = B * T * fh * fw, (fh // 5) * (fw // 9)
both = torch.randn(rows + B * T * nwin, 1536, device=dev)
qkv, kvp = both[:rows], both[rows:]
tab, nk = build_key_table(fh, fw, rolled_valid_index().tolist())
tab, nk = torch.from_numpy(tab).to(dev), torch.from_numpy(nk).to(dev)
out = torch.empty(rows, 512, device=dev)
ref = ops.focal_attention(qkv, kvp, tab, nk, B, T, fh, fw, waves=4).clone()
gflop = B * nwin * 4 * (45 * T) * (int(nk.float().mean().item()) * T) * 128 * 2 * 2 * 1e-9
for waves in (0, 2, 4, 12, 14, 22, 24, 32, 34):
    o = ops.focal_attention(qkv, kvp, tab, nk, B, T, fh, fw, out=out, waves=waves)
    diff = (o - ref).abs().max().item()
    iters = 10
    gr = torch.cuda.CUDAGraph(); st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        ops.focal_attention(qkv, kvp, tab, nk, B, T, fh, fw, out=out, waves=waves)
    torch.cuda.current_stream().wait_stream(st)
    with torch.cuda.graph(gr):
        for _ in range(iters):
            ops.focal_attention(qkv, kvp, tab, nk, B, T, fh, fw, out=out, waves=waves)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); torch.cuda.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / iters
    print("attention B=%d waves %d: %8.1f us  %5.1f TF (valid-key flops; diff %.1e)" % (B, waves, us, gflop / us * 1e3 / 1e3 * 1e3 / 1e3, diff), flush=True)
