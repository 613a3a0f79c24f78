#!/bin/bash
# quick loop for the wide-tile Winograd kernel: its tests + the per-layer microbenchmark.   bash tools/quick_w4.sh <tag>
O=gpurun_out/${1:-w4}; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_wino4.py -q -x ) > $O/pytest_w4.log 2>&1; grep -E "passed|failed|rror" $O/pytest_w4.log | tail -5
timeout 300 python tools/wino_bench.py 0 2464 2432 4432 > $O/wino_bench.txt 2>&1; tail -20 $O/wino_bench.txt
