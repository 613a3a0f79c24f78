mkdir -p gpurun_out/b1
timeout 900 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/b1/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/b1/pytest.log
python bench.py --no-cpu-baseline > gpurun_out/b1/bench.json 2> gpurun_out/b1/bench.err; python - <<'PY'
import json
j = json.load(open("gpurun_out/b1/bench.json"))
print("headline", j["value"], j["ms_per_step"], j["roofline"]["frac"])
for s in j.get("secondary", []):
    print(s.get("metric"), s.get("value"), s.get("ms_per_step"), s.get("roofline", {}).get("frac"), s.get("error"))
PY
