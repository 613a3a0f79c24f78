#!/bin/bash
# Round 5, sixth GPU call: the kernel table re-timed (the qkv Linear timed WITH its K / V-plane epilogue), then same-box A/Bs of the
# headline: default | K / V planes by the separate pass | the deformable conv on its two-K-group tile.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5f; mkdir -p $OUT
T0=$(date +%s); lap() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
timeout 700 python tools/make_tile_table.py gpurun_out/tiles > $OUT/tiles.log 2>&1; echo "tiles rc=$?"; tail -1 $OUT/tiles.log
if [ -s gpurun_out/tiles/tile_table.py ]; then cp gpurun_out/tiles/tile_table.py e2fgvi_amd/tile_table.py; fi; lap tiles
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python bench.py --no-cpu-baseline --no-secondary > $OUT/bench_$name.json 2> $OUT/bench_$name.err
  python -c "
import json
j=json.loads(open('$OUT/bench_$name.json').read().strip().splitlines()[-1])
print('$name', j['value'], j['ms_per_step'], 'dominant', j['roofline']['dominant_kernel']['avg_us'], j['config']['kernels'].get('transformer.*.qkv'))"
}
for rep in 1 2; do
  run default_$rep X=1
  run kv_pass_$rep E2FGVI_KV_EPILOGUE=0
  run dcn_tile4_$rep E2FGVI_DCN_TILE=4
done; lap ab
