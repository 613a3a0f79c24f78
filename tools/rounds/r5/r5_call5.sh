#!/bin/bash
# Round 5, fifth GPU call: DCN with scalar fragment offsets + three K groups by default (x3); the fusion layer in place; the
# qkv epilogue writing the attention's K / V planes.  Tests, then the headline A/B-free (compare with r5a: 805.6 on another box).
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5e; mkdir -p $OUT
T0=$(date +%s); lap() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
timeout 300 python tools/dcn_bench_x3.py > $OUT/dcn_x3_60x108.txt 2>&1; tail -9 $OUT/dcn_x3_60x108.txt; lap dcn
timeout 600 python -m pytest tests/test_gpu_x3.py tests/test_gpu_ops.py -q -x -p no:cacheprovider -k "mdcn or qkv_epilogue or focal_attention_x3 or planes" > $OUT/unit.log 2>&1; echo "unit rc=$?"; grep -a -E "passed|failed|Error" $OUT/unit.log | tail -3; lap unit
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_hazards.py -q -x -p no:cacheprovider > $OUT/model.log 2>&1; echo "model rc=$?"; grep -a -E "passed|failed|Error" $OUT/model.log | tail -3; lap model
for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-secondary > $OUT/bench_$rep.json 2> $OUT/bench_$rep.err
  python -c "
import json
j=json.loads(open('$OUT/bench_$rep.json').read().strip().splitlines()[-1])
print('run $rep', j['value'], j['ms_per_step'], 'dominant', j['roofline']['dominant_kernel']['avg_us'], j['library_sha16'])"
done; lap bench
timeout 200 python tools/layer_table.py --out $OUT/layer_table_fp32 > $OUT/layer_table.log 2>&1; grep -E "split3|qkv|dcn|fusion|copy" $OUT/layer_table_fp32.md | cut -c1-160; lap table
