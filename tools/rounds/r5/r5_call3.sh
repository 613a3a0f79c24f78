#!/bin/bash
# Round 5, third GPU call: deformable conv with the weights out of the LDS (B operand straight into registers), scalar source
# select, four-K-group tile -- parity tests and tile timings at the fp32 (60x108, split-operand) and bf16 (180x324) shapes.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5c; mkdir -p $OUT
T0=$(date +%s); lap() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
timeout 400 python -m pytest tests/test_gpu_ops.py tests/test_gpu_x3.py tests/test_gpu_bf16x.py -q -p no:cacheprovider -k "mdcn" > $OUT/mdcn_tests.log 2>&1; echo "mdcn tests rc=$?"; grep -a -E "passed|failed|Error" $OUT/mdcn_tests.log | tail -3; lap tests
timeout 200 python tools/dcn_bench_x3.py > $OUT/dcn_x3_60x108.txt 2>&1; cat $OUT/dcn_x3_60x108.txt | tail -10; lap x3
timeout 200 python tools/dcn_bench.py > $OUT/dcn_fp32_60x108.txt 2>&1; tail -12 $OUT/dcn_fp32_60x108.txt; lap fp32
DCN_TILES=1,2,4,5,6,7 timeout 300 python tools/dcn_bench_x.py > $OUT/dcn_bf16_180x324.txt 2>&1; tail -40 $OUT/dcn_bf16_180x324.txt; lap bf16
DCN_TILES=1,6,7 timeout 300 python tools/dcn_bench_x.py 270x486 > $OUT/dcn_bf16_270x486.txt 2>&1; tail -20 $OUT/dcn_bf16_270x486.txt; lap bf16_1080
