#!/bin/bash
# join position at the other configurations: 8 clips per forward, l_t = 5
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5j; mkdir -p $OUT
for j in 10 16 18; do
  E2FGVI_JOIN_AT=$j timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-dominant-probe --clips-per-gpu 8 --steps 10 > $OUT/b8_$j.json 2> $OUT/err.txt
  E2FGVI_JOIN_AT=$j timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-dominant-probe --lt 5 > $OUT/lt5_$j.json 2>> $OUT/err.txt
  python -c "
import json
a=json.loads(open('$OUT/b8_$j.json').read().strip().splitlines()[-1]); b=json.loads(open('$OUT/lt5_$j.json').read().strip().splitlines()[-1])
print('join at $j: 8 clips', a['value'], a['ms_per_step'], '| l_t=5', b['value'], b['ms_per_step'])"
done
