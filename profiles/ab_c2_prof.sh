#!/bin/bash
# rocprofv3 --stats of config C2 on the r02 build and on the current tree with SANAFE_PUSH=0 (same box): average kernel
# durations inside the running pipeline (not the flushed, event-timed steps).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp; export TMPDIR=/tmp
for spec in "r02:$ROOT/ab_r02:" "r03:$ROOT/ab_r03:SANAFE_PUSH=0" "r04:$ROOT:SANAFE_PUSH=0"; do
    name=${spec%%:*}; rest=${spec#*:}; dir=${rest%%:*}; envs=${rest#*:}
    out=$ROOT/gpurun_out/prof_c2_$name
    rm -rf "$out"; mkdir -p "$out"
    [ -n "$envs" ] && export $envs
    (cd "$dir" && rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 "$dir/bench.py" --workload c2 --steps 3000 --warmup 100 --no-cpu-baseline --timed-steps 0 > "$out/bench.json" 2> "$out/err.txt")
    [ -n "$envs" ] && unset ${envs%%=*}
    f=$(find "$out" -name "*kernel_stats.csv" | head -1)
    echo "== $name"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("   %-60s calls %6s avg_us %8.2f total_ms %8.2f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
