#!/bin/bash
# Round 5, seventh GPU call: the pipelined tiles of conv_bf16x (weights straight into registers, three A stages): parity, then
# tile timings on the 720p layer shapes against the tiles of the table.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5g; mkdir -p $OUT
T0=$(date +%s); lap() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
timeout 600 python -m pytest tests/test_gpu_bf16x.py -q -p no:cacheprovider -k "test_conv_bf16x" > $OUT/tests.log 2>&1; echo "tests rc=$?"; grep -a -E "passed|failed|Error|tile 5" $OUT/tests.log | tail -8; lap tests
timeout 600 python tools/bf16x_bench.py "" ${TILES:-1,51,4,54,6,56,7,2,52,5,55,11,17} > $OUT/bench.txt 2>&1; cat $OUT/bench.txt | cut -c1-420; lap bench
