#!/bin/bash
O=gpurun_out/c9; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_bf16x.py -q -x -s ) > $O/pytest_bf16x.log 2>&1
grep -E "passed|failed|rror|tile code" $O/pytest_bf16x.log | tail -8
( timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bf16x.py ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2
for i in 1 2; do timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1_$i.log 2>&1; tail -1 $O/bench_n1_$i.log | cut -c1-200; done
timeout 300 python tools/layer_table.py --out $O/layer_fp32_base > $O/layer1.log 2>&1; tail -1 $O/layer1.log
timeout 300 python tools/hq_run.py 720x1296 10 3 bf16 > $O/hq720_bf16.log 2>&1; tail -1 $O/hq720_bf16.log
timeout 300 python tools/hq_run.py 720x1296 10 3 fp32 > $O/hq720_fp32.log 2>&1; tail -1 $O/hq720_fp32.log
