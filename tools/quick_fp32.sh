#!/bin/bash
# quick loop for the fp32 headline: Winograd tests, A/B of the wide-tile rule, the bench line, the per-layer table.
#   bash tools/quick_fp32.sh <tag>
O=gpurun_out/${1:-qf}; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_wino4.py tests/test_gpu_ops.py -q -x -k "wino or dcn_post" ) > $O/pytest_wino.log 2>&1; grep -E "passed|failed|rror" $O/pytest_wino.log | tail -3
E2FGVI_WINO4_AUTO=0 timeout 300 python bench.py --no-cpu-baseline > $O/bench_w4off.json 2> $O/bench_w4off.err; tail -1 $O/bench_w4off.json | cut -c1-220
timeout 300 python bench.py --no-cpu-baseline > $O/bench_w4on.json 2> $O/bench_w4on.err; tail -1 $O/bench_w4on.json | cut -c1-220
WINO_ONLY_PROP=1 timeout 200 python tools/wino_bench.py 0 64 32 132 164 2464 > $O/wino_prop.txt 2>&1; grep "off.6" $O/wino_prop.txt
timeout 300 python tools/layer_table.py --out $O/layer_table_fp32_base 2>&1 | tail -1
