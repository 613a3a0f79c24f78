#!/bin/bash
# First GPU call of round 5: the kernel-level reproducer on the product library and on the two M0 variants.
#   CPU, before the call:   python tools/r4_variants.py build 32 64
#   gpurun --timeout 600 -- 'bash tools/r5_c4.sh'      -> gpurun_out/c4/
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out/c4
timeout 150 python tools/c4_repro.py 40 6 2>&1 | tee gpurun_out/c4/product.txt | tail -9
for v in 32 64; do
  [ -f tools/probe/libe2fgvi_x3v$v.so ] && E2FGVI_LIB=tools/probe/libe2fgvi_x3v$v.so timeout 150 python tools/c4_repro.py 40 6 2>&1 | tee gpurun_out/c4/v$v.txt | tail -9
done
