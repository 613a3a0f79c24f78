#!/bin/bash
# bash tools/traffic_by_stage.sh <tag> [traffic_by_stage.py args: --model --hw --t --precision]   -> gpurun_out/traffic_<tag>/by_stage.{md,json}
TAG=${1:-run}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/traffic_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o pmc -- python $REPO/tools/traffic_by_stage.py "$@" > $OUT/$C.log 2>&1 || tail -3 $OUT/$C.log
done
cd $REPO
python tools/traffic_by_stage.py --parse $OUT --out $OUT/by_stage --title "$TAG $*" "$@"
