#!/bin/bash
O=gpurun_out/c10; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_bf16x.py -q -x ) > $O/pytest_bf16x.log 2>&1; grep -E "passed|failed|rror" $O/pytest_bf16x.log | tail -3
timeout 400 python tools/bf16x_bench.py encoder.10,encoder.8,encoder.16,conv_offset.0,conv_offset.2,conv_offset.6,qkv,fc1,fc2,sc,"soft split" 1,6,7 > $O/bf16x_t7.log 2>&1; cat $O/bf16x_t7.log
timeout 300 python tools/hq_run.py 720x1296 10 3 bf16 > $O/hq720_bf16.log 2>&1; tail -1 $O/hq720_bf16.log
