#!/bin/bash
# SPyNet's two wide layers (32 -> 64, 64 -> 32, 7x7) of the upper levels on the split-operand GEMM instead of the fp32-MFMA halo kernel: headline A/B
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5y; mkdir -p $OUT
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-dominant-probe > $OUT/b.json 2> $OUT/b.err
  python -c "
import json
j=json.loads(open('$OUT/b.json').read().strip().splitlines()[-1])
print('$name', j['value'], j['ms_per_step'], j.get('parity'))" 2>/dev/null || tail -3 $OUT/b.err
}
for rep in 1 2; do
  run default_$rep X=1
  run spy_x3_lv5_$rep E2FGVI_SPY_X3=5,5,3
  run spy_x3_lv4_$rep E2FGVI_SPY_X3=4,5,3
  run spy_x3_lv3_$rep E2FGVI_SPY_X3=3,5,3
done
