#!/bin/bash
O=gpurun_out/c7; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_bf16x.py -q -x ) > $O/pytest_bf16x.log 2>&1
grep -E "passed|failed|rror" $O/pytest_bf16x.log | tail -8
( timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bf16x.py ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2
timeout 300 python tools/hq_run.py 720x1296 10 3 bf16 > $O/hq720_bf16.log 2>&1; tail -1 $O/hq720_bf16.log
timeout 300 python tools/layer_table.py --model e2fgvi_hq --hw 720x1296 --precision bf16 --out $O/layer_bf16_hq720 > $O/layer3.log 2>&1; tail -1 $O/layer3.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1.log 2>&1; tail -1 $O/bench_n1.log | cut -c1-200
bash tools/pmc_kernel.sh bf16x python $PWD/tools/bf16x_bench.py encoder.10,fc1,qkv,conv_offset.2 1,6 > $O/pmck.log 2>&1; cat gpurun_out/pmck_bf16x/summary.txt | cut -c1-220
