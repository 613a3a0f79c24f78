#!/bin/bash
# round-2 GPU call 1: full GPU test suite, cross-stream reproducer, bench lines, per-layer tables
O=gpurun_out/c1; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x --durations=15 ) > $O/pytest.log 2>&1
tail -5 $O/pytest.log
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/overlap_probe.hip -o /tmp/overlap_probe -ldl && timeout 300 /tmp/overlap_probe e2fgvi_amd/csrc/libe2fgvi_hip.so 40 ) > $O/overlap_probe.log 2>&1
tail -40 $O/overlap_probe.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_n1.log 2>&1; tail -1 $O/bench_n1.log
timeout 300 python bench.py --steps 20 --warmup 5 --force-dist --no-cpu-baseline > $O/bench_forcedist.log 2>&1; tail -1 $O/bench_forcedist.log
timeout 300 python bench.py --steps 10 --warmup 3 --clips-per-gpu 8 --no-cpu-baseline > $O/bench_b8.log 2>&1; tail -1 $O/bench_b8.log
timeout 300 python tools/layer_table.py --out $O/layer_fp32_base > $O/layer1.log 2>&1; tail -1 $O/layer1.log
timeout 300 python tools/layer_table.py --model e2fgvi_hq --hw 720x1296 --out $O/layer_fp32_hq720 > $O/layer2.log 2>&1; tail -1 $O/layer2.log
timeout 300 python tools/layer_table.py --model e2fgvi_hq --hw 720x1296 --precision bf16 --out $O/layer_bf16_hq720 > $O/layer3.log 2>&1; tail -1 $O/layer3.log
