#!/bin/bash
O=gpurun_out/c3; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_bf16x.py -q -x ) > $O/pytest_bf16x.log 2>&1
tail -15 $O/pytest_bf16x.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-330
E2FGVI_LIB=$PWD/e2fgvi_amd/csrc/libe2fgvi_hip_nopk.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_nopk_all.log 2>&1; tail -1 $O/bench_nopk_all.log | cut -c1-330
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_default2.log 2>&1; tail -1 $O/bench_default2.log | cut -c1-330
timeout 300 python tools/layer_table.py --out $O/layer_fp32_base > $O/layer1.log 2>&1; tail -1 $O/layer1.log
timeout 300 python tools/bf16x_bench.py > $O/bf16x_bench.log 2>&1; cat $O/bf16x_bench.log
( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/overlap_probe.hip -o /tmp/overlap_probe -ldl && timeout 400 /tmp/overlap_probe e2fgvi_amd/csrc/libe2fgvi_hip.so 30 1 ) > $O/overlap_probe.log 2>&1
grep -E "^V0" $O/overlap_probe.log | grep -v " 0 / "
grep -c "^V0" $O/overlap_probe.log
( timeout 600 python -m pytest tests -m gpu -q -x ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
