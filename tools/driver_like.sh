#!/bin/bash
# What the driver does at round end, in its order, on one box: GPU tests (-x), smoke(), the default bench line.
#   gpurun --timeout 1500 -- 'bash tools/driver_like.sh <tag>'    -> gpurun_out/driver_<tag>/
TAG=${1:-run}
OUT=gpurun_out/driver_$TAG
mkdir -p $OUT
timeout 1200 python -m pytest tests/ -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$? : $(grep -a -E "passed|failed" $OUT/pytest.log | tail -1)"
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > $OUT/smoke.log 2>&1; echo "smoke rc=$? : $(grep -a "smoke:" $OUT/smoke.log | tail -1)"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - "$OUT" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
print("headline %s %s ms frac %s alg %s | cpu %s" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"].get("frac_effective"), j.get("cpu_baseline", {}).get("value")))
for s in j.get("secondary", []):
    print("  ", s.get("metric"), s.get("value"), s.get("ms_per_step"), s.get("roofline", {}).get("frac"), s.get("error"))
PY
