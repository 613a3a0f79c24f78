#!/bin/bash
O=gpurun_out/c6; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_bf16x.py -q ) > $O/pytest_bf16x.log 2>&1
grep -E "passed|failed|rror" $O/pytest_bf16x.log | tail -8
timeout 300 python tools/hq_run.py 720x1296 10 3 bf16 > $O/hq720_bf16.log 2>&1; tail -1 $O/hq720_bf16.log
bash tools/profile.sh r02_fp32 > $O/profile_fp32.log 2>&1; tail -25 $O/profile_fp32.log | cut -c1-200
bash tools/profile.sh r02_hq720_bf16 --model e2fgvi_hq --hw 720x1296 --precision bf16 > $O/profile_hq.log 2>&1; tail -25 $O/profile_hq.log | cut -c1-200
bash tools/pmc.sh r02 > $O/pmc.log 2>&1; tail -3 $O/pmc.log | cut -c1-600
