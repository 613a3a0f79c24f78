#!/bin/bash
# SPyNet layers as candidates of the split-operand GEMM: table re-timed, flow / overlap / hazard tests, headline
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5aa; mkdir -p $OUT
T0=$(date +%s); lap() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-dominant-probe > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "
import json
j=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1])
print('$name', j['value'], j['ms_per_step'], {k: v for k, v in j['config']['kernels'].items() if 'spynet' in k})"
}
run before_1 X=1; run before_2 X=1
timeout 700 python tools/make_tile_table.py gpurun_out/tiles > $OUT/tiles.log 2>&1; echo "tiles rc=$?"; tail -1 $OUT/tiles.log
if [ -s gpurun_out/tiles/tile_table.py ]; then cp gpurun_out/tiles/tile_table.py e2fgvi_amd/tile_table.py; fi; lap tiles
run after_1 X=1; run after_2 X=1; lap bench
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_hazards.py -x -q -m gpu -k "stage_flows or stream_overlap or hazards or beside or golden or full_size" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$? : $(grep -a -E 'passed|failed' $OUT/pytest.log | tail -1)"; lap tests
