#!/bin/bash
# ms/frame of every bench workload (no CPU baseline): tools/bench_all.sh [workloads...]
cd "${GRAFT_REPO_ROOT:-/root/repo}"
wl=${@:-sponza sponza4k dragon dragon80k glass spheres}
for w in $wl; do
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --workload $w | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['config']
print('%-10s %.4f ms/frame %8.1f Mrays/s  one-call %.4f (defer off %.4f)  gpu-event %.4f' % ('$w', d['ms_per_step'], d['value'], c['ms_per_step_one_call_per_frame'] or 0, c.get('ms_per_step_one_call_per_frame_defer_off') or 0, c['gpu_event_ms_per_step']))" || exit 1
done
