#!/bin/bash
# frames -> 4-channel pixels by the one-thread-per-pixel kernel; stream priorities (A/B); the table re-timed with tile 8 of conv_bf16x
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5u; mkdir -p $OUT
T0=$(date +%s); lap() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
timeout 300 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "layout" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$? : $(grep -a -E 'passed|failed' $OUT/pytest.log | tail -1)"
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-dominant-probe > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "
import json
j=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1])
print('$name', j['value'], j['ms_per_step'], j['config']['kernels'].get('transformer.*.qkv'))"
}
for rep in 1 2; do
  run default_$rep X=1
  run main_high_$rep E2FGVI_CAPTURE_PRIORITY=-1
  run side_high_$rep E2FGVI_SIDE_PRIORITY=-1
done; lap prio
timeout 700 python tools/make_tile_table.py gpurun_out/tiles > $OUT/tiles.log 2>&1; echo "tiles rc=$?"; tail -1 $OUT/tiles.log
if [ -s gpurun_out/tiles/tile_table.py ]; then cp gpurun_out/tiles/tile_table.py e2fgvi_amd/tile_table.py; fi; lap tiles
for rep in 1 2; do run retimed_$rep X=1; done; lap retimed
