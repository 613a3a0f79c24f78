#!/bin/bash
# Everything the round's numbers come from, in one GPU-box call (about 6 minutes):
#   bash tools/measure_all.sh <tag>      -> gpurun_out/<tag>/
O=gpurun_out/${1:-all}; mkdir -p $O
( timeout 900 python -m pytest tests -q -x -m gpu ) > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
timeout 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json
timeout 400 python bench.py --model e2fgvi_hq --hw 720x1296 --precision bf16 > $O/bench_hq720_bf16.json 2> $O/bench_hq720_bf16.err; tail -1 $O/bench_hq720_bf16.json
timeout 300 python tools/hq_run.py 720x1296 10 3 fp32 > $O/hq720_fp32.log 2>&1; tail -1 $O/hq720_fp32.log
timeout 300 python tools/hq_run.py 1080x1944 20 2 bf16 > $O/hq1080_bf16.log 2>&1; tail -1 $O/hq1080_bf16.log
timeout 300 python tools/layer_table.py --out $O/layer_table_fp32_base 2>&1 | tail -1
timeout 300 python tools/layer_table.py --model e2fgvi_hq --hw 720x1296 --precision bf16 --out $O/layer_table_hq720_bf16 2>&1 | tail -1
