#!/bin/bash
# Per-kernel times (HIP events) of config C2 on the r02 / r03 / current builds, same box (see ab_c2.sh).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
run() {
    (cd "$2" && env $3 timeout -k 10 120 python3 bench.py --workload c2 --steps 3000 --warmup 100 --no-cpu-baseline --traffic none --timed-steps 200 2>/dev/null |
        python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
nk=r.get('neuron_kernel') or {}; dk=r.get('deliver_kernel') or {}
n = nk.get('avg_launch_ms', r['avg_launch_ms'] if r['kernel']=='neuron_kernel' else None)
de = dk.get('avg_launch_ms', r['avg_launch_ms'] if r['kernel']!='neuron_kernel' else None)
print('$1', round(d['value']), 'us/step', round(1e3*d['ms_per_step'],2), 'neuron_us', round(1e3*n,2), 'deliver_us', round(1e3*de,2), 'reduce_us', round(1e3*r['reduce_kernel_avg_ms'],2), d['config']['device_layout'])")
}
run r02 "$ROOT/ab_r02" ""
run r03_nopush "$ROOT/ab_r03" "SANAFE_PUSH=0"
run r04_nopush "$ROOT" "SANAFE_PUSH=0"
run r04 "$ROOT" ""
