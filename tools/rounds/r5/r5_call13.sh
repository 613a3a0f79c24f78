#!/bin/bash
# staggered 8-wave split-operand Winograd kernel (tile codes + 2000): parity, then device time against the lockstep shapes
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5s; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_x3.py -x -q -m gpu -k "winograd_x3" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$? : $(grep -a -E 'passed|failed' $OUT/pytest.log | tail -1)"
timeout 600 python tools/x3_bench.py conv_offset.0,conv_offset.2,backbone.0,backbone.b0,hq.conv_offset.2,conv_offset.6,encoder.6,encoder.16 99 2>&1 | tee $OUT/x3_bench.txt | cut -c1-400
