#!/bin/bash
# per-kernel PMC summary of an arbitrary command (two passes: SQ / LDS / MFMA, then L2 hit rate).
#   bash tools/pmc_kernel.sh <tag> <command...>    -> gpurun_out/pmck_<tag>/summary.txt
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmck_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/p1 -o pmc -- "$@" > $OUT/run1.log 2>&1 || true
timeout 300 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum --kernel-trace --output-format csv -d $OUT/p2 -o pmc -- "$@" > $OUT/run2.log 2>&1 || true
python - "$OUT" <<'PY' | tee $OUT/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(collections.Counter)
cnt = collections.Counter()
for sub in ("p1", "p2"):
    for f in glob.glob(out + "/" + sub + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            k = k[:100]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                cnt[k] += 1
print("%-100s %5s %12s %9s %9s %9s %9s %9s %8s" % ("kernel", "calls", "gui_cycles", "mfma/gui", "wait/wave", "winst/wv", "ldsconf", "ldsact/g", "L2hit"))
for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0))[:30]:
    g = c.get("GRBM_GUI_ACTIVE", 0) or 1
    wv = c.get("SQ_WAVE_CYCLES", 0) or 1
    hit, miss = c.get("TCC_HIT_sum", 0), c.get("TCC_MISS_sum", 0)
    print("%-100s %5d %12.0f %9.2f %9.3f %9.3f %9.3f %9.2f %8.3f" % (k, cnt[k], g, c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / g, c.get("SQ_WAIT_ANY", 0) / wv,
          c.get("SQ_WAIT_INST_ANY", 0) / wv, c.get("SQ_LDS_BANK_CONFLICT", 0) / (c.get("SQ_LDS_IDX_ACTIVE", 0) or 1), c.get("SQ_LDS_IDX_ACTIVE", 0) / g,
          hit / (hit + miss) if hit + miss else 0))
PY
