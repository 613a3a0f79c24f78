#!/bin/bash
# rocprofv3 kernel trace + stats of the benchmark forward.  Usage (on the GPU box, from the repo root):
#   bash tools/profile.sh <tag> [bench args...]      -> gpurun_out/prof_<tag>/
set -e
TAG=${1:-run}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export E2FGVI_AUTOTUNE=0     # no tile-tuning launches inside the profiled run (static tile choice)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o prof -- python $REPO/bench.py --no-cpu-baseline --no-graph --steps 5 --warmup 2 "$@" > $OUT/bench.log 2>&1 || true
tail -1 $OUT/bench.log
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
# drop the (large) per-dispatch trace, keep the stats
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
python - "$OUT/kernel_stats.csv" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-90s %8s %10s %8s" % ("kernel", "calls", "total_ms", "pct"))
for r in rows[:40]:
    print("%-90s %8s %10.3f %7.2f%%" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot))
PY
