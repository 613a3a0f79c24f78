#!/bin/bash
# Round 4, one GPU-box call: what the driver does (tests, smoke, default bench line) + the profiles kept under profiles/r04_*
#   gpurun --timeout 1500 -- 'bash tools/r4_final.sh'    -> gpurun_out/r4_final/
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd $REPO
O=gpurun_out/r4_final; mkdir -p $O
timeout 1200 python -m pytest tests/ -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$? : $(grep -a -E "passed|failed" $O/pytest.log | tail -1)"
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > $O/smoke.log 2>&1; echo "smoke rc=$? : $(grep -a "smoke:" $O/smoke.log | tail -1)"
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - "$O" <<'PY'
import json, sys
j = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
d = j["roofline"]["dominant_kernel"]
print("headline %s %s ms frac %s | parity %s | dominant %s %s us | weights %s | cpu %s" % (j["value"], j["ms_per_step"], j["roofline"]["frac"],
      j.get("parity", {}).get("max_abs"), d["kernel"][:40], d["avg_us"], j.get("weight_memory_gb"), j.get("cpu_baseline", {}).get("value")))
for s in j.get("secondary", []):
    print("  ", s.get("metric"), s.get("value"), s.get("ms_per_step"), s.get("error"))
PY
bash tools/profile.sh r04_fp32 --no-secondary > $O/profile.log 2>&1; tail -2 $O/profile.log | cut -c1-200
TR=$(find gpurun_out/prof_r04_fp32 -name "*kernel_trace.csv" | head -1)
[ -n "$TR" ] && python tools/timeline.py $TR 15 > $O/timeline.txt 2>&1
bash tools/pmc_dom.sh 46064 "conv_wino_x3w<64>" > $O/pmc_dom.log 2>&1; cp gpurun_out/pmc_dom/traffic.json $O/dominant_kernel_traffic.json; grep hbm_bytes $O/dominant_kernel_traffic.json
timeout 300 python tools/layer_table.py --out $O/layer_table_fp32 2>&1 | tail -1
find gpurun_out -name "*kernel_trace.csv" -size +8M -delete; find gpurun_out/pmc_dom -name "*.csv" -size +4M -delete
