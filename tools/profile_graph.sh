#!/bin/bash
# rocprofv3 kernel trace of the benchmark forward REPLAYED FROM ITS HIP GRAPH (what bench.py times): shows how the side stream
# (SPyNet) and the main stream overlap inside the graph.   bash tools/profile_graph.sh <tag> [bench args]  -> gpurun_out/profg_<tag>/
TAG=${1:-run}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/profg_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $REPO/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 2 "$@" > $OUT/warm.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT -o prof -- python $REPO/bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 2 "$@" > $OUT/bench.log 2>&1 || true
tail -1 $OUT/bench.log | cut -c1-200
python $REPO/tools/timeline.py $(find $OUT -name "*kernel_trace.csv" | head -1) 0 > $OUT/timeline.txt 2>&1
tail -2 $OUT/timeline.txt
