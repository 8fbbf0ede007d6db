#!/bin/bash
# Collects the rocprofv3 evidence for one bench configuration on the GPU box:
#   profiles/collect.sh <name> [bench.py arguments...]
# Three separate passes (kernel trace + stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE: the two TCC counters do not
# fit one pass, MI355X_MICROARCH.md "rocprofv3 PMC slots"), then profiles/summarize.py condenses them into
# gpurun_out/prof_<name>/ which is copied by hand into profiles/<name>/.
set -eo pipefail
NAME=$1
shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/prof_$NAME
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" "$@" > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
echo "[collect] trace pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --timed-steps 0 > "$OUT/bench_under_fetch.json" 2> "$OUT/fetch.err"
echo "[collect] FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --timed-steps 0 > "$OUT/bench_under_write.json" 2> "$OUT/write.err"
echo "[collect] WRITE_SIZE pass done"
python3 "$ROOT/profiles/summarize.py" "$OUT"
