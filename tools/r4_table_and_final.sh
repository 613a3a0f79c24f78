#!/bin/bash
# One GPU-box call: regenerate the kernel table with the tree's flags, install it, then tools/r4_final.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
mkdir -p gpurun_out/tiles
E2FGVI_TUNE_REPS=3 timeout 900 python tools/make_tile_table.py gpurun_out/tiles 2>&1 | tail -1
cp gpurun_out/tiles/tile_table.py e2fgvi_amd/tile_table.py
bash tools/r4_final.sh
for i in 1 2 3 4; do timeout 300 python -m pytest tests/test_gpu_model.py -q -k "full_size_properties or overlap" 2>&1 | tail -1; done
