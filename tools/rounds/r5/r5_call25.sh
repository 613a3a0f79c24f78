#!/bin/bash
# propagation: from step 2 on the second-order part of conv_offset.0 (cond_n2 + flows, on top of the current-frame part) a step ahead on the
# side stream (E2FGVI_PROP_AHEAD, default 1): parity, table re-timed with the new layers, same-box A/B
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5ad; mkdir -p $OUT
T0=$(date +%s); lap() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "stage_propagation or end_to_end or golden or stream_overlap" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$? : $(grep -a -E 'passed|failed' $OUT/pytest.log | tail -1)"; lap tests
timeout 700 python tools/make_tile_table.py gpurun_out/tiles > $OUT/tiles.log 2>&1; echo "tiles rc=$?"; tail -1 $OUT/tiles.log
if [ -s gpurun_out/tiles/tile_table.py ]; then cp gpurun_out/tiles/tile_table.py e2fgvi_amd/tile_table.py; fi; lap tiles
run() { # name, args, env...
  local name=$1; local args=$2; shift; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-dominant-probe $args > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "
import json
j=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1])
print('$name', j['value'], j['ms_per_step'], {k: v for k, v in j['config']['kernels'].items() if 'order part' in k})" 2>/dev/null || tail -2 $OUT/bench_$name.err
}
for rep in 1 2; do
  run ahead_$rep "" X=1
  run no_ahead_$rep "" E2FGVI_PROP_AHEAD=0
  run whole_$rep "" E2FGVI_PROP_SPLIT=0
  run lt5_ahead_$rep "--lt 5" X=1
  run lt5_no_ahead_$rep "--lt 5" E2FGVI_PROP_AHEAD=0
done; lap ab
bash tools/profile_graph.sh r05_fp32 2>&1 | tail -1; lap timeline
