#!/bin/bash
# rocprofv3 kernel stats of the HQ model at a given size:  bash tools/profile_hq.sh <tag> 720x1296 10
TAG=${1:-hq}; HW=${2:-720x1296}; T=${3:-10}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o prof -- python $REPO/tools/hq_run.py $HW $T 1 > $OUT/run.log 2>&1 || true
tail -1 $OUT/run.log
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
python - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-90s %8s %10s %8s" % ("kernel", "calls", "total_ms", "pct"))
for r in rows[:24]:
    print("%-90s %8s %10.3f %7.2f%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
