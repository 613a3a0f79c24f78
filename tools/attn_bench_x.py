"""Micro-benchmark of the bf16 fused focal attention at the e2fgvi_hq shapes (720x1296 T=10: 60x108 tokens; 1080x1944 T=20):
every kernel variant (1 = round 2's register-staged kernel; 10 QB + NW = the LDS-DMA kernel, NW waves of QB x 32 queries per
workgroup) timed on the same tensors, its output compared with variant 1's (which the parity tests pin to the oracle).
    python tools/attn_bench_x.py [fhxfw] [T] [variants, comma separated]          (E2FGVI_LIB=<other build> for A/B runs)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from e2fgvi_amd import ops
from e2fgvi_amd.engine import build_key_table
from e2fgvi_amd.synth import rolled_valid_index

dev = torch.device("cuda:0")
fh, fw = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "60x108").split("x"))
T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
variants = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "1,12,14,18,22,24,28").split(",")]
B = 1
rows, nwin = B * T * fh * fw, (fh // 5) * (fw // 9)
torch.manual_seed(0)
both = (torch.randn(rows + B * T * nwin, 1536, device=dev) * 0.5).bfloat16()
qkv, kvp = both[:rows], both[rows:]
tab, nk = build_key_table(fh, fw, rolled_valid_index().tolist())
tab, nk = torch.from_numpy(tab).to(dev), torch.from_numpy(nk).to(dev)
gflop = B * 4 * (45 * T) * 128 * 2 * 2 * float(nk.float().sum().item()) * T * 1e-9
ref = None
for var in variants:
    out = torch.zeros(rows, 512, device=dev, dtype=torch.bfloat16)
    try:
        best = 1e30
        for rep in range(3):
            ops.focal_attention_bf16(qkv, kvp, tab, nk, B, T, fh, fw, out=out, variant=var)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.focal_attention_bf16(qkv, kvp, tab, nk, B, T, fh, fw, out=out, variant=var)
            e1.record(); torch.cuda.synchronize()
            best = min(best, 1e3 * e0.elapsed_time(e1) / 10)
    except Exception as e:
        print("variant %d: %s" % (var, str(e).splitlines()[0]), flush=True)
        continue
    if ref is None:
        ref = out.float()
    d = (out.float() - ref).abs()
    print("attention bf16 %dx%d T=%d variant %2d: %8.1f us  %6.1f TF/s (valid-key flops)  max |out - variant %d| %.3e (rms of out %.3e, "
          "mismatching > 2 bf16 ulp: %d of %d)" % (fh, fw, T, var, best, gflop / best * 1e3, variants[0], d.max().item(),
                                                  ref.pow(2).mean().sqrt().item(),
                                                  int((d > 2.0 ** -7 * ref.abs() + 1e-2 * ref.pow(2).mean().sqrt()).sum().item()), out.numel()),
          flush=True)
