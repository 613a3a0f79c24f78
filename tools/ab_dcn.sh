#!/bin/bash
# Same-box A/B of deformable-conv variants INSIDE the forward (profiles/r02_dcn_sampler.txt).  E2FGVI_LIB points the binding at
# another build of the library: build the variant, copy libe2fgvi_hip.so to gpurun_tmp/libe2fgvi_<tag>.so (git-ignored, travels
# to the GPU box), rebuild the tree, then:   bash tools/ab_dcn.sh <tag>
TAG=${1:-olddcn}
run() { python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
HQ="--model e2fgvi_hq --hw 720x1296 --precision bf16"
for rep in 1 2; do
  echo "bf16 tree, 8x8 blocks:";  E2FGVI_DCN_BLOCKS=1 run $HQ
  echo "bf16 tree, row tiles:";   E2FGVI_DCN_BLOCKS=0 run $HQ
  echo "bf16 $TAG:";              E2FGVI_LIB=$PWD/gpurun_tmp/libe2fgvi_$TAG.so run $HQ
done
echo "fp32 tree:"; run
echo "fp32 $TAG:"; E2FGVI_LIB=$PWD/gpurun_tmp/libe2fgvi_$TAG.so run
echo "fp32 tree:"; run
