# A/B of an environment switch inside the bf16 720p forward:  bash tools/ab_env.sh VAR [reps]
VAR=$1; REPS=${2:-2}
run() { python bench.py --no-cpu-baseline --steps 20 --warmup 3 --model e2fgvi_hq --hw 720x1296 --precision bf16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'])"; }
for rep in $(seq $REPS); do for v in 1 0; do echo "$VAR=$v:"; env $VAR=$v bash -c "$(declare -f run); run"; done; done
