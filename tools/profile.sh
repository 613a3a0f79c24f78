#!/bin/bash
# rocprofv3 kernel trace + stats of the benchmark forward, with the tile choices the benchmark uses (the checked-in
# table e2fgvi_amd/tile_table.py: no tuning launches inside).
# Usage (on the GPU box, from the repo root):   bash tools/profile.sh <tag> [bench args...]   -> gpurun_out/prof_<tag>/
TAG=${1:-run}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py --no-cpu-baseline --steps 10 --warmup 3 "$@" > $OUT/bench_graph.log 2>&1
tail -1 $OUT/bench_graph.log > $OUT/bench_line.json
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o prof -- python $REPO/bench.py --no-cpu-baseline --no-graph --steps 5 --warmup 2 "$@" > $OUT/bench.log 2>&1 || true
tail -1 $OUT/bench.log | cut -c1-300
find $OUT -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
python - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, sys, glob, collections, re
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "/kernel_stats.csv")))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("# rocprofv3 --kernel-trace --stats of: bench.py --no-graph --steps 5 --warmup 2 (9 eager forwards: engine build, FLOP trace, 2 warm-up, 5 timed; the fp32 headline also runs the dominant-kernel probe: 21 launches of encoder.layers.10)")
print("%-96s %8s %10s %10s %8s" % ("kernel", "calls", "total_ms", "avg_us", "pct"))
for r in rows[:40]:
    print("%-96s %8s %10.3f %10.2f %7.2f%%" % (r["Name"][:96], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                            100 * float(r["TotalDurationNs"]) / tot))
# the same trace split by launch geometry: one row per (kernel, grid) = per layer shape, so that a single layer's average
# (e.g. the dominant kernel on encoder.layers.10, which bench.py's roofline.dominant_kernel quotes) can be read off directly
tr = glob.glob(out + "/**/*kernel_trace.csv", recursive=True)
if tr:
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(tr[0])):
        n = r["Kernel_Name"]
        m = re.search(r"([a-z_0-9]+_kernel<[^>]*>|[a-z_0-9]+_kernel)", n)
        k = (m.group(1) if m else n[:60], r["Grid_Size_X"] + "x" + r.get("Grid_Size_Y", "1"), r["Workgroup_Size_X"])
        a = acc[k]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print()
    print("# by launch geometry (kernel, grid threads X x Y, workgroup): the 30 largest")
    print("%-60s %14s %6s %8s %10s %10s" % ("kernel", "grid", "wg", "calls", "total_ms", "avg_us"))
    for k, (c, us) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:30]:
        print("%-60s %14s %6s %8d %10.3f %10.2f" % (k[0][:60], k[1], k[2], c, us / 1e3, us / c))
PY
find $OUT -name "*kernel_trace.csv" -size +20M -delete || true
