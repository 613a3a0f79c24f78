#!/bin/bash
# Round 6, final measurement pass on one box (what the driver does, then the profiles the bench line and DESIGN.md quote):
#   GPU suite (-x), smoke(), default bench line, rocprof summaries + PMC traffic of configs[1] / [3] / [4], the dominant kernel's
#   traffic, traffic by stage, per-layer tables, the graph-replay timeline, the forced-gather 8-clip step.
#   gpurun --timeout 3000 -- 'bash tools/r6_final.sh'      -> gpurun_out/r6z/, gpurun_out/{prof,pmc,traffic}_r06*/
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r6z; mkdir -p $OUT
T0=$(date +%s); lap() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
timeout 1500 python -m pytest tests/ -x -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$? : $(grep -a -E 'passed|failed' $OUT/pytest.log | tail -1)"; lap suite
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > $OUT/smoke.log 2>&1; echo "smoke rc=$? : $(grep -a 'smoke:' $OUT/smoke.log | tail -1)"; lap smoke
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - "$OUT" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
    print("headline %s %s ms (sequential %s) frac %s useful %s eff %s | cpu %s | parity %s | traffic %s" % (j["value"], j["ms_per_step"], j.get("sequential", {}).get("value"), j["roofline"]["frac"], j["roofline"].get("frac_useful"), j["roofline"].get("frac_effective"), j.get("cpu_baseline", {}).get("value"), {k: (v if not isinstance(v, dict) else v["max_abs"]) for k, v in j.get("parity", {}).items() if k not in ("vs",)}, j["roofline"].get("traffic")))
    d = j["roofline"].get("dominant_kernel", {}); print("dominant", d.get("avg_us"), d.get("kernel"), d.get("frac"), d.get("frac_useful"), d.get("traffic"))
    for s in j.get("secondary", []):
        print("  ", s.get("metric"), s.get("value"), s.get("ms_per_step"), "seq", s.get("sequential", {}).get("value"), s.get("roofline", {}).get("frac"), {k: v for k, v in (s.get("parity") or {}).items() if k in ("max_abs", "rms_of_difference_over_rms")}, s.get("error"))
except Exception as e:
    print("bench line unreadable:", e)
PY
lap bench
bash tools/profile.sh r06_fp32 --no-secondary 2>&1 | tail -1; lap prof_fp32
bash tools/pmc.sh r06 2>&1 | tail -1 | cut -c1-200; lap pmc_fp32
bash tools/pmc_dom.sh 46064 "conv_wino_x3w<64>" > /dev/null 2>&1; cut -c1-300 gpurun_out/pmc_dom/traffic.json | head -12; lap pmc_dom
bash tools/traffic_by_stage.sh r06 > $OUT/traffic_by_stage.log 2>&1; tail -3 $OUT/traffic_by_stage.log | cut -c1-200; lap traffic_by_stage
HQ7="--model e2fgvi_hq --hw 720x1296 --precision bf16 --no-secondary"
bash tools/profile.sh r06_hq720_bf16 $HQ7 2>&1 | tail -1; lap prof_hq720
bash tools/pmc.sh r06_hq720_bf16 $HQ7 2>&1 | tail -1 | cut -c1-200; lap pmc_hq720
bash tools/traffic_by_stage.sh r06_hq720_bf16 --model e2fgvi_hq --hw 720x1296 --precision bf16 > $OUT/traffic_by_stage_hq720.log 2>&1; tail -2 $OUT/traffic_by_stage_hq720.log | cut -c1-200; lap traffic_hq720
HQ="--model e2fgvi_hq --hw 1080x1944 --t 20 --precision bf16 --no-secondary"
bash tools/profile.sh r06_hq1080_bf16 $HQ 2>&1 | tail -1; lap prof_hq1080
bash tools/pmc.sh r06_hq1080_bf16 $HQ 2>&1 | tail -1 | cut -c1-200; lap pmc_hq1080
timeout 200 python tools/layer_table.py --out $OUT/layer_table_fp32 > $OUT/layer_table_fp32.log 2>&1; lap table_fp32
timeout 300 python tools/layer_table.py --model e2fgvi_hq --hw 720x1296 --precision bf16 --out $OUT/layer_table_hq720_bf16 > $OUT/layer_table_hq720.log 2>&1; lap table_hq720
bash tools/profile_graph.sh r06_fp32 2>&1 | tail -1; lap timeline
timeout 300 python bench.py --force-dist --clips-per-gpu 8 --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_8clips_forced_gather.json 2> $OUT/bench_8clips.err; python - "$OUT" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1] + "/bench_8clips_forced_gather.json").read().strip().splitlines()[-1])
    print("8 clips, forced gather (one rank): %s frames/s %s ms | same work, no collective: %s" % (j["value"], j["ms_per_step"], j["single_gpu_same_work"]))
except Exception as e:
    print("8-clip line unreadable:", e)
PY
lap gather
timeout 300 python bench.py --gpus 1 --launcher --clips-per-gpu 8 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_self_launch.json 2> $OUT/bench_self_launch.err; echo "self-launch rc=$? $(tail -1 $OUT/bench_self_launch.json | cut -c1-160)"; lap self_launch
# keep what travels back small: the raw counter CSVs of the PMC passes stay on the box
find gpurun_out -name "*counter_collection.csv" -size +2M -delete; find gpurun_out -name "*kernel_trace.csv" -size +2M -delete
du -sh gpurun_out | tail -1
