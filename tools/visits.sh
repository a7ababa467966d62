cd $GRAFT_REPO_ROOT
for w in sponza sponza4k dragon dragon80k glass; do
  python3 bench.py --no-cpu-baseline --no-single-call --workload $w --steps 32 --warmup 16 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']
print('$w', d['ms_per_step'], d['value'], r['node4q_visits_per_ray'], r['tri48_tests_per_ray'], r['kernel_ms_per_frame'], r['shade_ms_per_frame'], r['frac'], r['kernel_own_frac'])"
done
