#!/bin/bash
# rocprofv3 kernel trace + stats of the benchmark forward, with the SAME tile choices the benchmark uses: a first,
# un-profiled run writes the autotuner's decisions to a file, the profiled run reads them (no tuning launches inside).
# Usage (on the GPU box, from the repo root):   bash tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
TAG=${1:-run}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export E2FGVI_TUNE_FILE=$OUT/tune.txt
rm -f $E2FGVI_TUNE_FILE
python $REPO/bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" > $OUT/bench_graph.log 2>&1
tail -1 $OUT/bench_graph.log > $OUT/bench_line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o prof -- python $REPO/bench.py --no-cpu-baseline --no-graph --steps 5 --warmup 2 "$@" > $OUT/bench.log 2>&1 || true
tail -1 $OUT/bench.log | cut -c1-300
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
python - "$OUT/kernel_stats.csv" <<'PY' | tee $OUT/summary.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# rocprofv3 --kernel-trace --stats of: bench.py --no-graph --steps 5 --warmup 2 (9 eager forwards: engine build, FLOP trace, 2 warm-up, 5 timed)")
print("%-96s %8s %10s %10s %8s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
for r in rows[:40]:
    print("%-96s %8s %10.3f %10.2f %7.2f%%" % (r["Name"][:96], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                            100 * float(r["TotalDurationNs"]) / tot))
PY
