#!/bin/bash
# round-2 GPU call 2: library without packed-fp32 VALU, Winograd block autotune, split propagation convs
O=gpurun_out/c2; mkdir -p $O
export TMPDIR=/tmp
( time timeout 900 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_n1.log 2>&1; tail -1 $O/bench_n1.log | cut -c1-400
timeout 300 python tools/layer_table.py --out $O/layer_fp32_base > $O/layer1.log 2>&1; tail -1 $O/layer1.log
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/overlap_probe.hip -o /tmp/overlap_probe -ldl && timeout 400 /tmp/overlap_probe e2fgvi_amd/csrc/libe2fgvi_hip.so 30 1 ) > $O/overlap_probe.log 2>&1
grep -v "0 / " $O/overlap_probe.log | tail -40
timeout 200 python tools/overlap_probe.py 100 > $O/overlap_probe_py.log 2>&1; tail -12 $O/overlap_probe_py.log
