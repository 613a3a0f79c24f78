mkdir -p gpurun_out/s1
timeout 600 python -m pytest tests/test_gpu_bf16x.py -q -m gpu -p no:cacheprovider -k "softcomp_gather or bf16_path" 2>&1 | tail -12
for g in 1 0 1 0; do E2FGVI_SC_GATHER=$g python bench.py --model e2fgvi_hq --hw 720x1296 --precision bf16 --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SC_GATHER=$g', j['value'], j['ms_per_step'])" | tee -a gpurun_out/s1/ab.txt; done
