#!/bin/bash
# Same-box A/B of config C2: the r02 build, the r03 build (default and SANAFE_PUSH=0) and the current
# tree, each `bench.py --workload c2 --steps 3000`, interleaved three times.  ab_r02/ and ab_r03/ are `git archive`s of the
# round-end commits built in place (git-ignored; they travel to the GPU box with the snapshot).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/ab_c2.txt
: > "$OUT"
run() { # name dir [env]
    (cd "$2" && env $3 timeout -k 10 120 python3 bench.py --workload c2 --steps 3000 --warmup 100 --no-cpu-baseline --timed-steps 0 2>/dev/null |
        python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(1e3*d['ms_per_step'],2))") >> "$OUT"
}
for rep in 1 2 3; do
    run r02 "$ROOT/ab_r02" ""
    run r03 "$ROOT/ab_r03" ""
    run r03_nopush "$ROOT/ab_r03" "SANAFE_PUSH=0"
    run r04 "$ROOT" ""
    run r04_nopush "$ROOT" "SANAFE_PUSH=0"
done
cat "$OUT"
