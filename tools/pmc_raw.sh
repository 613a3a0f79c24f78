#!/bin/bash
# raw per-kernel counter sums of an arbitrary command, one rocprofv3 pass per counter group (groups separated by '/').
#   bash tools/pmc_raw.sh <tag> "<ctr ctr .../ctr ctr ...>" <command...>    -> gpurun_out/pmcr_<tag>/summary.txt
TAG=$1; GROUPS_=$2; shift; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcr_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
IFS='/' read -ra GR <<< "$GROUPS_"
for g in "${GR[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $g --kernel-trace --output-format csv -d $OUT/p$i -o pmc -- "$@" > $OUT/run$i.log 2>&1 || tail -3 $OUT/run$i.log
done
python - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(collections.Counter)
calls = collections.defaultdict(collections.Counter)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:90]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", kv[1].get("GRBM_GUI_ACTIVE", 0)))[:6]:
    print(k)
    for name in sorted(c):
        print("    %-40s %16.0f   (%d dispatches)" % (name, c[name], calls[k][name]))
PY
