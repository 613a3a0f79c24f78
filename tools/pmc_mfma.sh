#!/bin/bash
# MFMA-pipe utilisation per kernel from rocprofv3 PMC counters (own pass, kernel-trace only).
#   bash tools/pmc_mfma.sh <tag>   -> gpurun_out/pmc_mfma_<tag>/summary.txt
TAG=${1:-run}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_mfma_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (kernel selection is the checked-in table: no tuning launches inside the counted forwards)
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p -o pmc -- python $REPO/bench.py --no-cpu-baseline --no-secondary --no-dominant-probe --no-graph --steps 2 --warmup 1 > $OUT/run.log 2>&1 || true
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
files = glob.glob(out + "/p/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.Counter())
cnt = collections.Counter()
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = k[k.find("::") + 2:][:70] if "anonymous" in k else k[:70]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            cnt[k] += 1
            try:                                     # device time of the dispatch: the clock the chip held = GUI-active cycles / ns
                acc[k]["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            except (KeyError, ValueError):
                pass
rows = []
for k, c in acc.items():
    g = c.get("GRBM_GUI_ACTIVE", 0.0)
    m = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    rows.append((g, k, m, c.get("SQ_BUSY_CYCLES", 0.0), cnt[k], c.get("ns", 0.0)))
rows.sort(reverse=True)
tot_g = sum(r[0] for r in rows)
lines = ["%-72s %6s %14s %16s %9s %9s" % ("kernel", "calls", "GRBM_GUI_ACTIVE", "MFMA_BUSY_CYCLES", "mfma/gui", "GHz")]
for g, k, m, b, n, ns in rows[:30]:
    lines.append("%-72s %6d %14.0f %16.0f %9.2f %9.2f" % (k, n, g, m, m / g if g else 0, g / ns if ns else 0))
lines.append("sum GRBM_GUI_ACTIVE %.0f, sum MFMA busy %.0f, ratio %.2f" % (tot_g, sum(r[2] for r in rows), sum(r[2] for r in rows) / tot_g))
lines.append("normalisation: SQ_VALU_MFMA_BUSY_CYCLES is summed over the sampled SQ instances; divide mfma/gui by the value a pure-MFMA kernel reaches to read it as utilisation")
open(out + "/summary.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
