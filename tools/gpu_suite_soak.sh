#!/bin/bash
# Run the whole GPU suite several times on one box, each run with another PYTHONHASHSEED and another E2FGVI_TEST_SEED
# (tests/util.py: shifts every test's data), never with -x, and record how far below its bound every comparison sat.
#   gpurun --timeout 1500 -- 'bash tools/gpu_suite_soak.sh 5'
# Output: gpurun_out/soak/run<i>.log, margins<i>.tsv (what, measured, allowed, ratio), summary.txt (worst ratios).
N=${1:-5}
OUT=gpurun_out/soak
mkdir -p $OUT
rm -f $OUT/margins*.tsv
rc_all=0
for i in $(seq 0 $((N - 1))); do
  PYTHONHASHSEED=$((i * 7919 + 1)) E2FGVI_TEST_SEED=$i E2FGVI_TEST_MARGINS=$OUT/margins$i.tsv \
    timeout 1200 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/run$i.log 2>&1
  rc=$?
  [ $rc -ne 0 ] && rc_all=$rc
  echo "run $i (PYTHONHASHSEED=$((i * 7919 + 1)) E2FGVI_TEST_SEED=$i): rc=$rc  $(tail -1 $OUT/run$i.log)"
done | tee $OUT/summary.txt
python tools/margin_summary.py $OUT/margins*.tsv | tee -a $OUT/summary.txt
exit $rc_all
