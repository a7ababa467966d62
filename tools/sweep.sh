#!/bin/bash
# developer sweep of the fast-pipeline knobs on the bench workload
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for cfg in "$@"; do
  env $cfg python3 bench.py --no-cpu-baseline --steps 32 --warmup 4 2>&1 | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        j = json.loads(l); r = j['roofline']
        print('$cfg', 'ms', j['ms_per_step'], 'Mrays', j['value'], 'trace', r['kernel_ms_per_frame'], 'shade', r['shade_ms_per_frame'])
"
done
