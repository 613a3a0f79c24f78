#!/bin/bash
# Copies what DESIGN.md / README.md / bench.py quote from the scratch output of tools/r6_final.sh (gpurun_out/) into profiles/r06_*.
#   bash tools/r6_collect.sh <file with the stdout of the gpurun call>
cd "$(dirname "$0")/.."; G=gpurun_out; P=profiles
for t in fp32 hq720_bf16 hq1080_bf16; do
  cp $G/prof_r06_$t/summary.txt $P/r06_${t}_summary.txt; cp $G/prof_r06_$t/kernel_stats.csv $P/r06_${t}_kernel_stats.csv; cp $G/prof_r06_$t/bench_line.json $P/r06_${t}_bench_line.json
done
cp $G/pmc_r06/traffic.json $P/r06_hbm_traffic.json; cp $G/pmc_r06_hq720_bf16/traffic.json $P/r06_hbm_traffic_hq720_bf16.json; cp $G/pmc_r06_hq1080_bf16/traffic.json $P/r06_hbm_traffic_hq1080_bf16.json
cp $G/pmc_dom/traffic.json $P/r06_dominant_kernel_traffic.json
cp $G/traffic_r06/by_stage.md $P/r06_traffic_by_stage.md; cp $G/traffic_r06/by_stage.json $P/r06_traffic_by_stage.json
cp $G/traffic_r06_hq720_bf16/by_stage.md $P/r06_traffic_by_stage_hq720_bf16.md; cp $G/traffic_r06_hq720_bf16/by_stage.json $P/r06_traffic_by_stage_hq720_bf16.json
for t in fp32 hq720_bf16; do cp $G/r6z/layer_table_$t.md $P/r06_layer_table_$t.md; cp $G/r6z/layer_table_$t.json $P/r06_layer_table_$t.json; done
cp $G/profg_r06_fp32/timeline.txt $P/r06_timeline.txt
cp $G/r6z/bench.json $P/r06_bench_default_with_secondary.json
cp $G/r6z/bench_8clips_forced_gather.json $P/r06_bench_8clips_forced_gather.json; cp $G/r6z/bench_self_launch.json $P/r06_bench_self_launch_one_rank.json
[ -n "$1" ] && grep -a -v "amdgpu.ids" "$1" | cut -c1-400 > $P/r06_driver_like_run.txt
git status --short $P | head -40
