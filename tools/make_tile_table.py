"""Generates e2fgvi_amd/tile_table.py, the checked-in kernel-selection table (ops.py: deterministic selection).
Runs the BASELINE configurations with E2FGVI_AUTOTUNE=1 (timing-based selection, every candidate of every layer timed on the
actual call, best of E2FGVI_TUNE_REPS x the usual launches, SPyNet NOT overlapped with the encoder so that nothing else runs
beside a timed launch) in child processes -- the split-operand alternatives on and off -- and merges their decisions.
    GPU:  python tools/make_tile_table.py gpurun_out/tiles        -> gpurun_out/tiles/tile_table.py (+ raw decision files)
    then: cp gpurun_out/tiles/tile_table.py e2fgvi_amd/tile_table.py"""
import ast, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (model, precision, H, W, clips, t, l_t)
CONFIGS = [("e2fgvi", "fp32", 240, 432, 1, 10, 10), ("e2fgvi", "fp32", 240, 432, 1, 10, 5), ("e2fgvi", "fp32", 240, 432, 8, 10, 10),
           ("e2fgvi", "fp32", 240, 432, 1, 5, 5), ("e2fgvi", "fp32", 240, 432, 2, 10, 10), ("e2fgvi", "fp32", 240, 432, 1, 18, 11),
           ("e2fgvi_hq", "fp32", 240, 432, 1, 10, 10), ("e2fgvi_hq", "fp32", 360, 648, 1, 10, 10),
           ("e2fgvi_hq", "bf16", 240, 432, 1, 10, 10), ("e2fgvi_hq", "bf16", 720, 1296, 1, 10, 10), ("e2fgvi_hq", "bf16", 1080, 1944, 1, 20, 20),
           ("e2fgvi", "bf16", 240, 432, 1, 10, 10)]

if len(sys.argv) > 2 and sys.argv[1] == "--child":
    import importlib
    import torch
    from e2fgvi_amd import ops
    from e2fgvi_amd.synth import synth_clip, synth_state_dict
    assert ops.AUTOTUNE
    dev = torch.device("cuda:0")
    for model, precision, H, W, b, t, lt in CONFIGS:
        if precision == "bf16" and not ops.X3_ENABLED:
            continue                                         # the bf16 path has no split-operand alternatives: once is enough
        net = importlib.import_module("model." + model).InpaintGenerator()
        net.load_state_dict(synth_state_dict(model, "default", 0))
        net = net.to(dev).eval()
        net.precision = precision
        net.engine().overlap_flows = False
        x = synth_clip(b, t, H, W, seed=0, smooth=False)[0].to(dev)
        n0 = len(ops._TUNED)
        with torch.no_grad():
            net(x, lt)
            net(x, lt)
        torch.cuda.synchronize()
        print("%s %s %dx%d b=%d t=%d l_t=%d: %d new decisions" % (model, precision, H, W, b, t, lt, len(ops._TUNED) - n0), flush=True)
        del net, x
        torch.cuda.empty_cache()
    sys.exit(0)

out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "tiles")
os.makedirs(out, exist_ok=True)
tiles = {}
for x3 in ("1", "0"):
    raw = os.path.join(out, "raw_x3_%s.txt" % x3)
    if os.path.exists(raw):
        os.remove(raw)
    env = dict(os.environ, E2FGVI_AUTOTUNE="1", E2FGVI_TUNE_FILE=raw, E2FGVI_TUNE_REPS=os.environ.get("E2FGVI_TUNE_REPS", "3"), E2FGVI_X3=x3)
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", x3], env=env)
    for line in open(raw):
        k, v = ast.literal_eval(line)
        tiles[k] = v
from e2fgvi_amd import ops
with open(os.path.join(out, "tile_table.py"), "w") as fh:
    fh.write('"""Kernel-selection table of e2fgvi_amd.ops (deterministic selection): (layer geometry, size class, ...) -> tile code.\n'
             'GENERATED on an MI355X by tools/make_tile_table.py from timed runs of the BASELINE configurations -- do not edit by hand;\n'
             'regenerate after a kernel change.  Key layouts: ops.PackedConv.__call__ / ops.PackedConvX.__call__."""\n')
    fh.write("TABLE_FORMAT = %d\nTILES = {\n" % ops.TABLE_FORMAT)
    for k in sorted(tiles, key=repr):
        fh.write("    %r: %r,\n" % (k, tiles[k]))
    fh.write("}\n")
print("wrote %s: %d decisions" % (os.path.join(out, "tile_table.py"), len(tiles)))
