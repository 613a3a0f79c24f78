#!/bin/bash
O=gpurun_out/c8; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_bf16x.py -q -x ) > $O/pytest_bf16x.log 2>&1
grep -E "passed|failed|rror" $O/pytest_bf16x.log | tail -8
timeout 400 python tools/bf16x_bench.py > $O/bf16x_bench.log 2>&1; cat $O/bf16x_bench.log
timeout 300 python tools/hq_run.py 720x1296 10 3 bf16 > $O/hq720_bf16.log 2>&1; tail -1 $O/hq720_bf16.log
timeout 300 python tools/layer_table.py --model e2fgvi_hq --hw 720x1296 --precision bf16 --out $O/layer_bf16_hq720 > $O/layer3.log 2>&1; tail -1 $O/layer3.log
