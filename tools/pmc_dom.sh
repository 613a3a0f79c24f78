#!/bin/bash
# HBM-side traffic of the dominant kernel (encoder.layers.10), per launch:
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes -> gpurun_out/pmc_dom/traffic.json
#   bash tools/pmc_dom.sh [tile code] [kernel tag]     default: 40164 "conv_wino_x3<1,64>" (the split-operand Winograd kernel the
#   fp32 engine picks for this layer since round 3); 0 "conv_wino4<F(2x4),64>" = round 2's fp32 kernel
TILE=${1:-40164}; TAG=${2:-"conv_wino_x3<1,64>"}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_dom; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/$c -o pmc -- python $REPO/tools/wino_one.py 5 $TILE > $OUT/run_$c.log 2>&1 || tail -3 $OUT/run_$c.log
done
cd $REPO
python - "$OUT" "$TILE" "$TAG" <<'PY' | tee $OUT/traffic.json
import csv, glob, sys, json, collections, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT") or os.getcwd())
out = sys.argv[1]
acc = collections.defaultdict(collections.Counter); calls = collections.Counter()
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_wino" in r["Kernel_Name"] and "pack" not in r["Kernel_Name"]:
            acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "FETCH_SIZE":
                calls[r["Kernel_Name"]] += 1
k = max(acc, key=lambda n: acc[n]["FETCH_SIZE"])
n = calls[k]
fetch, write = acc[k]["FETCH_SIZE"] * 1024 * 2 / n, acc[k]["WRITE_SIZE"] * 1024 / n
tile, tag = sys.argv[2], sys.argv[3]
# input + output + the transformed weights (fp32 F(2x4): 24 positions x 4 bytes; split-operand F(2x2): 16 positions x 3 bf16 planes)
alg = (10 * 60 * 108 * 640 + 10 * 60 * 108 * 512) * 4 + (512 * 320 * 16 * 6 if "x3" in tag else 512 * 320 * 24 * 4)
print(json.dumps({"kernel": k[:120] + " on encoder.layers.10 (3x3 640->512 g2, 10x60x108), tools/wino_one.py 5 " + tile, "kernel_tag": tag, "launches": n,
                  "FETCH_SIZE_KiB_total": acc[k]["FETCH_SIZE"], "WRITE_SIZE_KiB_total": acc[k]["WRITE_SIZE"],
                  "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write,
                  "algorithmic_bytes_per_launch": alg, "library_sha16": __import__("e2fgvi_amd.lib", fromlist=["x"]).library_key(),
                  "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc_dom.sh), KiB units, read side doubled per the gfx950 calibration of MI355X_MICROARCH.md; fabric-side (includes Infinity-Cache hits)"}, indent=1))
PY
