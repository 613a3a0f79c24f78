#!/bin/bash
O=gpurun_out/${1:-qm}; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_bf16x.py -q -x ) > $O/pytest_ops.log 2>&1; grep -E "passed|failed|rror" $O/pytest_ops.log | tail -3
timeout 300 python tools/hq_run.py 720x1296 10 3 bf16 > $O/hq720_bf16.log 2>&1; tail -1 $O/hq720_bf16.log
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | cut -c1-200
timeout 300 python tools/layer_table.py --model e2fgvi_hq --hw 720x1296 --precision bf16 --out $O/layer_table_hq720_bf16 2>&1 | tail -1
