#!/bin/bash
# quick loop for the fp32 headline: Winograd / DCN-epilogue tests, the bench line.   bash tools/quick_fp32.sh <tag>
O=gpurun_out/${1:-qf}; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_wino4.py tests/test_gpu_ops.py -q -x -k "wino or dcn_post" ) > $O/pytest_wino.log 2>&1; grep -E "passed|failed|rror" $O/pytest_wino.log | tail -3
timeout 300 python bench.py --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err; tail -1 $O/bench_default.json | cut -c1-220
