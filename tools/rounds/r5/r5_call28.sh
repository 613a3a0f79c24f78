#!/bin/bash
# the propagation split in the bf16 data path: parity (bf16 golden fixtures, bf16 stream-overlap identity), table re-timed, same-box A/B at 720p / 1080p
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5ae; mkdir -p $OUT
T0=$(date +%s); lap() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_bf16x.py -x -q -m gpu -k "bf16 and (golden or overlap or path or end_to_end)" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$? : $(grep -a -E 'passed|failed' $OUT/pytest.log | tail -1)"; lap tests
timeout 700 python tools/make_tile_table.py gpurun_out/tiles > $OUT/tiles.log 2>&1; echo "tiles rc=$?"; tail -1 $OUT/tiles.log
if [ -s gpurun_out/tiles/tile_table.py ]; then cp gpurun_out/tiles/tile_table.py e2fgvi_amd/tile_table.py; fi; lap tiles
run() { # name, args, env...
  local name=$1; local args=$2; shift; shift
  env "$@" timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-dominant-probe $args > $OUT/b_$name.json 2> $OUT/b_$name.err
  python -c "
import json
j=json.loads(open('$OUT/b_$name.json').read().strip().splitlines()[-1])
print('$name', j['value'], j['ms_per_step'], j.get('peak_memory_gb'))" 2>/dev/null || tail -2 $OUT/b_$name.err
}
HQ7="--model e2fgvi_hq --hw 720x1296 --precision bf16"; HQ10="--model e2fgvi_hq --hw 1080x1944 --t 20 --precision bf16 --steps 8"
for rep in 1 2; do
  run hq720_split_$rep "$HQ7" X=1
  run hq720_whole_$rep "$HQ7" E2FGVI_PROP_SPLIT=0
  run hq1080_split_$rep "$HQ10" X=1
  run hq1080_whole_$rep "$HQ10" E2FGVI_PROP_SPLIT=0
done
run fp32_split "" X=1; run fp32_whole "" E2FGVI_PROP_SPLIT=0; lap ab
