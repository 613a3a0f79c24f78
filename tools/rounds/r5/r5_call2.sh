#!/bin/bash
# Round 5, second GPU call: the round-to-nearest split (csrc/common.h e2_split2: v_cvt_pk_bf16_f32 + v_dot2c_f32_bf16) against the
# truncating one of rounds 3-4 -- exactness probe on the chip, the split-operand kernel tests, same-box A/B of the headline and of
# single layers.  CPU, before the call:  python -c 'from e2fgvi_amd import build; build.build_variant("trunc", "-DE2_SPLIT_RNE=0")'
#   gpurun --timeout 900 -- 'bash tools/rounds/r5/r5_call2.sh'      -> gpurun_out/r5b/
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5b; mkdir -p $OUT
T0=$(date +%s); lap() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
for O in 1 0; do
  (timeout 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -DOPAQUE=$O tools/probe/split_probe.hip -o /tmp/split_probe$O 2>/dev/null && timeout 60 /tmp/split_probe$O) > $OUT/split_probe_opaque$O.txt 2>&1; echo "OPAQUE=$O: $(tail -2 $OUT/split_probe_opaque$O.txt | tr '\n' ' ')"
done; lap probe
timeout 600 python -m pytest tests/test_gpu_x3.py tests/test_gpu_hazards.py -q -p no:cacheprovider > $OUT/x3_tests.log 2>&1; echo "x3 tests rc=$?"; grep -a -E "passed|failed" $OUT/x3_tests.log | tail -1; lap x3tests
timeout 600 python -m pytest tests/test_gpu_model.py -q -x -p no:cacheprovider -k "full_size or golden or end_to_end or stage_" > $OUT/model_tests.log 2>&1; echo "model tests rc=$?"; grep -a -E "passed|failed" $OUT/model_tests.log | tail -1; lap modeltests
TR=$REPO/e2fgvi_amd/csrc/libe2fgvi_hip_trunc.so
for rep in 1 2; do
  for which in rne trunc; do
    if [ $which = trunc ]; then export E2FGVI_LIB=$TR; else unset E2FGVI_LIB; fi
    timeout 300 python bench.py --no-cpu-baseline --no-secondary > $OUT/bench_${which}_$rep.json 2> $OUT/bench_${which}_$rep.err
    python -c "
import json,sys
j=json.loads(open('$OUT/bench_${which}_$rep.json').read().strip().splitlines()[-1])
print('$which $rep', j['value'], j['ms_per_step'], 'dominant', j['roofline']['dominant_kernel']['avg_us'], j['library_sha16'])"
  done
done; unset E2FGVI_LIB; lap ab_headline
LAYERS=encoder.10,encoder.8,encoder.16,decoder.4,decoder.0,conv_offset.0,conv_offset.2,conv_offset.6,backbone.0,fc1,qkv,proj,sc,ss
for which in rne trunc; do
  if [ $which = trunc ]; then export E2FGVI_LIB=$TR; else unset E2FGVI_LIB; fi
  timeout 300 python tools/x3_bench.py $LAYERS > $OUT/x3_bench_$which.txt 2>&1; tail -30 $OUT/x3_bench_$which.txt
done; unset E2FGVI_LIB; lap x3bench
timeout 200 python tools/attn_bench_x3.py > $OUT/attn_rne.txt 2>&1; tail -4 $OUT/attn_rne.txt
E2FGVI_LIB=$TR timeout 200 python tools/attn_bench_x3.py > $OUT/attn_trunc.txt 2>&1; tail -4 $OUT/attn_trunc.txt; lap attn
