#!/bin/bash
# conv_bf16x tile 8 (256 x 192, 8 waves of 64 x 96): parity in all three modes, then device time against tiles 6 / 7 on the token GEMMs
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5t; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_bf16x.py tests/test_gpu_x3.py -x -q -m gpu -k "test_conv_bf16x or test_conv_x3 or qkv_epilogue or test_conv_f32x" -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$? : $(grep -a -E 'passed|failed' $OUT/pytest.log | tail -1)"
timeout 600 python tools/x3_bench.py qkv,fc1,fc2,proj,sc,ss,fusion 4,6,7,8 2>&1 | tee $OUT/x3_bench.txt | cut -c1-330
