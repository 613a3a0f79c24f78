#!/bin/bash
# Round 5: where should the SPyNet stream join the main one?  In front of encoder.layers.10 (default) or behind .10 / .12 / .14 / .16
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5i; mkdir -p $OUT
for rep in 1 2; do for j in 10 12 14 16 18; do
  E2FGVI_JOIN_AT=$j timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-dominant-probe > $OUT/bench_$j_$rep.json 2> $OUT/err.txt
  python -c "
import json
j=json.loads(open('$OUT/bench_$j_$rep.json').read().strip().splitlines()[-1]); print('join at $j run $rep:', j['value'], j['ms_per_step'])"
done; done
