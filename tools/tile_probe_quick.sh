cd $GRAFT_REPO_ROOT
for cfg in "OKRT_BENCH_TILE=8,0,8" "OKRT_BENCH_TILE=8,0,4" "OKRT_BENCH_TILE=8,0,2" "X=1"; do
  env $cfg python3 bench.py --no-cpu-baseline --no-single-call --steps 256 --warmup 64 | python3 -c "import json,sys; d=json.loads(sys.stdin.readline()); print('$cfg', 'ms/step', d['ms_per_step'])"
done
