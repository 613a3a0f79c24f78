#!/bin/bash
# fork x join sweep of the SPyNet branch on the headline (frames/s, ms per forward); two passes over the grid
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5w; mkdir -p $OUT
for rep in 1 2; do
 for f in 0 2 4 6 8; do
  for j in 14 16 18; do
    E2FGVI_FORK_AT=$f E2FGVI_JOIN_AT=$j timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-dominant-probe > $OUT/b.json 2> $OUT/b.err
    python -c "
import json
j=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('fork $f join $j rep $rep:', j['value'], j['ms_per_step'])"
  done
 done
done
