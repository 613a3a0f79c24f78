#!/bin/bash
# quick loop for bf16-path work: kernel tests + the 720p forward time.   bash tools/quick_bf16.sh <tag>
O=gpurun_out/${1:-q}; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_bf16x.py -q -x ) > $O/pytest_bf16x.log 2>&1; grep -E "passed|failed|rror" $O/pytest_bf16x.log | tail -5
timeout 300 python tools/hq_run.py 720x1296 10 3 bf16 > $O/hq720_bf16.log 2>&1; tail -1 $O/hq720_bf16.log
