#!/bin/bash
# PMC counters per kernel for an arbitrary command (own pass, kernel-trace only; rocprofv3).
#   bash tools/pmc_cmd.sh <tag> "<counters>" <command...>   -> gpurun_out/pmc_<tag>/summary.txt
TAG=$1; CTRS=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d $OUT/p -o pmc -- "$@" > $OUT/run.log 2>&1 || true
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(collections.Counter); cnt = collections.Counter()
for f in glob.glob(out + "/p/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k[k.find("::") + 2:][:60] if "anonymous" in k else k[:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
names = sorted({c for v in acc.values() for c in v})
lines = ["%-62s %6s " % ("kernel", "calls") + " ".join("%22s" % n for n in names)]
for k, c in sorted(acc.items(), key=lambda kv: -max(kv[1].values()))[:12]:
    lines.append("%-62s %6d " % (k, cnt[(k, names[0])]) + " ".join("%22.0f" % c.get(n, 0) for n in names))
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
