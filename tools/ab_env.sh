# A/B of an environment switch on the built library, interleaved: tools/ab_env.sh VAR "bench args" [reps]
R=$PWD; VAR=$1; ARGS=$2; N=${3:-3}
run() { timeout -k 10 150 python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-one-shot $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_per_step',{}); print('$1', 'ms', round(d['ms_per_step'],2), 'ext', round(k.get('ms_extend',0),2), 'shd', round(k.get('ms_shade',0),2))"; }
for i in $(seq $N); do run "default"; export $VAR=1; run "$VAR=1"; unset $VAR; done
