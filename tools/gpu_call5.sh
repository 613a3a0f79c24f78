#!/bin/bash
O=gpurun_out/c5; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_bf16x.py -q -s ) > $O/pytest_bf16x.log 2>&1
grep -E "passed|failed|rror|bf16 path|flows:" $O/pytest_bf16x.log | tail -25
( timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bf16x.py ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 300 python tools/hq_run.py 720x1296 10 3 bf16 > $O/hq720_bf16.log 2>&1; tail -1 $O/hq720_bf16.log
timeout 300 python tools/layer_table.py --model e2fgvi_hq --hw 720x1296 --precision bf16 --out $O/layer_bf16_hq720 > $O/layer3.log 2>&1; tail -1 $O/layer3.log
timeout 300 python tools/hq_run.py 1080x1944 20 2 bf16 > $O/hq1080_bf16.log 2>&1; tail -1 $O/hq1080_bf16.log
timeout 600 python bench.py --model e2fgvi_hq --hw 720x1296 --precision bf16 --steps 10 --warmup 3 > $O/bench_hq720_bf16.log 2>&1; tail -1 $O/bench_hq720_bf16.log | cut -c1-1500
