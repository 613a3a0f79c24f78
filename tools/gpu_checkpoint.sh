#!/bin/bash
# GPU suite, smoke, the default bench line, and a soak of the bench invocation that found DESIGN.md C8 (graph replay after earlier
# graphs of the process were destroyed):   gpurun --timeout 2400 -- 'bash tools/gpu_checkpoint.sh [soak runs]'   -> gpurun_out/r6c/
OUT=gpurun_out/r6c; mkdir -p $OUT; N=${1:-6}
T0=$(date +%s); lap() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$? : $(grep -a -E 'passed|failed' $OUT/pytest.log | tail -1)"; lap suite
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > $OUT/smoke.log 2>&1; echo "smoke rc=$? : $(grep -a 'smoke:' $OUT/smoke.log | tail -1)"; lap smoke
for i in $(seq 1 $N); do timeout 600 python -X faulthandler bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/soak$i.json 2> $OUT/soak$i.err; echo "soak $i rc=$? $(python -c "import json,sys; j=json.loads(open('$OUT/soak$i.json').read().strip().splitlines()[-1]); print(j['value'], [s.get('value', s.get('error')) for s in j['secondary']])" 2>&1 | tail -1)"; done; lap soak
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; lap bench
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r6c/bench.json").read().strip().splitlines()[-1])
print("headline", j["value"], j["ms_per_step"], "seq", j.get("sequential", {}).get("value"), "frac", j["roofline"]["frac"], "parity", {k: (v if not isinstance(v, dict) else v.get("max_abs")) for k, v in j.get("parity", {}).items() if k in ("default", "stress", "peaked")})
for s in j.get("secondary", []):
    print("  ", s.get("metric"), s.get("value"), "seq", s.get("sequential", {}).get("value"), (s.get("parity") or {}).get("max_abs"))
PY
