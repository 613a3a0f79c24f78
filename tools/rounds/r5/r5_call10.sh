#!/bin/bash
# Round 5: the GPU suite under three data sets / hash seeds without -x (margins of every comparison), then the default bench line on the
# final tree -- now that profiles/ holds the PMC traffic of this very library, the line quotes it.
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; mkdir -p gpurun_out/r5h
bash tools/gpu_suite_soak.sh 3 2>&1 | tail -40 | cut -c1-220
timeout 900 python bench.py > gpurun_out/r5h/bench.json 2> gpurun_out/r5h/bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r5h/bench.json").read().strip().splitlines()[-1])
print("headline", j["value"], j["ms_per_step"], "traffic", j["roofline"]["traffic"], j["roofline"].get("traffic_note", "")[:120])
print("dominant", {k: v for k, v in j["roofline"]["dominant_kernel"].items() if k in ("avg_us", "frac", "traffic", "traffic_note")})
print("parity", {k: (v if not isinstance(v, dict) else v["max_abs"]) for k, v in j["parity"].items() if k != "vs"})
for s in j.get("secondary", []): print("  ", s.get("metric"), s.get("value"), s.get("ms_per_step"), s.get("roofline", {}).get("frac"), s.get("error"))
PY
