#!/bin/bash
# Round 5, first GPU call: (1) probe of the cheaper exact split, (2) the C4 aggressor tests, (3) the kernel table re-timed without
# round 4's per-layer gate, (4) the whole GPU suite with it, (5) the default bench line, (6) rocprof summaries + PMC traffic of the
# fp32 headline and of configs[4] (e2fgvi_hq 1080x1944 T=20 bf16), (7) per-layer tables.  Every step has its own timeout.
#   gpurun --timeout 1800 -- 'bash tools/rounds/r5/r5_call1.sh'      -> gpurun_out/r5a/
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO; OUT=gpurun_out/r5a; mkdir -p $OUT
T0=$(date +%s); lap() { echo "== $1: $(( $(date +%s) - T0 )) s"; }
(timeout 120 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/probe/split_probe.hip -o /tmp/split_probe 2>/dev/null && timeout 60 /tmp/split_probe) > $OUT/split_probe.txt 2>&1; tail -3 $OUT/split_probe.txt; lap probe
timeout 600 python -m pytest tests/test_gpu_hazards.py -q -p no:cacheprovider > $OUT/hazards.log 2>&1; echo "hazards rc=$?"; tail -5 $OUT/hazards.log; lap hazards
timeout 700 python tools/make_tile_table.py gpurun_out/tiles > $OUT/tiles.log 2>&1; echo "tiles rc=$?"; tail -2 $OUT/tiles.log
if [ -s gpurun_out/tiles/tile_table.py ]; then cp gpurun_out/tiles/tile_table.py e2fgvi_amd/tile_table.py; fi; lap tiles
timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log; lap suite
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
python - "$OUT" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1] + "/bench.json").read().strip().splitlines()[-1])
    print("headline %s %s ms frac %s alg %s | cpu %s | parity %s" % (j["value"], j["ms_per_step"], j["roofline"]["frac"], j["roofline"]["frac_algorithmic"], j.get("cpu_baseline", {}).get("value"), {k: v for k, v in j.get("parity", {}).items() if k != "vs"}))
    print("dominant", j["roofline"].get("dominant_kernel", {}).get("avg_us"), j["roofline"].get("dominant_kernel", {}).get("kernel"))
    for s in j.get("secondary", []):
        print("  ", s.get("metric"), s.get("value"), s.get("ms_per_step"), s.get("roofline", {}).get("frac"), s.get("error"))
except Exception as e:
    print("bench line unreadable:", e)
PY
lap bench
bash tools/profile.sh r05_fp32 --no-secondary 2>&1 | tail -2; lap prof_fp32
bash tools/pmc.sh r05 2>&1 | tail -1 | cut -c1-400; lap pmc_fp32
HQ="--model e2fgvi_hq --hw 1080x1944 --t 20 --precision bf16 --no-secondary"
bash tools/profile.sh r05_hq1080_bf16 $HQ 2>&1 | tail -2; lap prof_hq1080
bash tools/pmc.sh r05_hq1080_bf16 $HQ 2>&1 | tail -1 | cut -c1-400; lap pmc_hq1080
timeout 200 python tools/layer_table.py --out gpurun_out/r5a/layer_table_fp32 > $OUT/layer_table_fp32.log 2>&1; lap table_fp32
timeout 300 python tools/layer_table.py --model e2fgvi_hq --hw 720x1296 --precision bf16 --out gpurun_out/r5a/layer_table_hq720_bf16 > $OUT/layer_table_hq720.log 2>&1; lap table_hq720
bash tools/profile_graph.sh r05_fp32 2>&1 | tail -2; lap timeline
