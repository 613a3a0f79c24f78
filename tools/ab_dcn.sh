run() { python bench.py --no-cpu-baseline --steps 20 --warmup 3 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
HQ="--model e2fgvi_hq --hw 720x1296 --precision bf16"
echo "bf16 new blocks=1:"; E2FGVI_DCN_BLOCKS=1 run $HQ
echo "bf16 new blocks=0:"; E2FGVI_DCN_BLOCKS=0 run $HQ
echo "bf16 old dcn:";      E2FGVI_LIB=$PWD/gpurun_tmp/libe2fgvi_olddcn.so run $HQ
echo "bf16 new blocks=1:"; E2FGVI_DCN_BLOCKS=1 run $HQ
echo "bf16 new blocks=0:"; E2FGVI_DCN_BLOCKS=0 run $HQ
echo "fp32 new:"; run
echo "fp32 old dcn:"; E2FGVI_LIB=$PWD/gpurun_tmp/libe2fgvi_olddcn.so run
echo "fp32 new:"; run
