#!/bin/bash
O=gpurun_out/c4; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_bf16x.py -q ) > $O/pytest_bf16x.log 2>&1
grep -E "passed|failed|Error|error|assert|bf16 path" $O/pytest_bf16x.log | tail -25
( timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_bf16x.py ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
for p in fp32 bf16; do timeout 300 python tools/hq_run.py 720x1296 10 3 $p > $O/hq720_$p.log 2>&1; tail -1 $O/hq720_$p.log; done
timeout 300 python tools/layer_table.py --model e2fgvi_hq --hw 720x1296 --precision bf16 --out $O/layer_bf16_hq720 > $O/layer3.log 2>&1; tail -1 $O/layer3.log
timeout 300 python tools/hq_run.py 1080x1944 20 2 bf16 > $O/hq1080_bf16.log 2>&1; tail -1 $O/hq1080_bf16.log
