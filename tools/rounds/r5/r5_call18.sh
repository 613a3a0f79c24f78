#!/bin/bash
# the bf16 HQ configurations on the table with / without conv_bf16x tile 8 (same box, alternating)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5x; mkdir -p $OUT
run() { # name, args, env...
  local name=$1; local args=$2; shift; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-dominant-probe $args > $OUT/b.json 2> $OUT/b.err
  python -c "
import json
j=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('$name', j['value'], j['ms_per_step'])"
}
HQ7="--model e2fgvi_hq --hw 720x1296 --precision bf16"; HQ10="--model e2fgvi_hq --hw 1080x1944 --t 20 --precision bf16 --steps 8"
for rep in 1 2; do
  run hq720_new_$rep "$HQ7" X=1
  run hq720_old_$rep "$HQ7" E2FGVI_TILE_TABLE=tools/tables/tile_table_before_tile8.py
  run hq1080_new_$rep "$HQ10" X=1
  run hq1080_old_$rep "$HQ10" E2FGVI_TILE_TABLE=tools/tables/tile_table_before_tile8.py
  run fp32_new_$rep "" X=1
  run fp32_old_$rep "" E2FGVI_TILE_TABLE=tools/tables/tile_table_before_tile8.py
done
