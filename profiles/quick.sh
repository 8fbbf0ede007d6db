#!/bin/bash
# Quick look at kernel durations: profiles/quick.sh <name> [bench.py args]  -> gpurun_out/q_<name>.txt
set -o pipefail
NAME=$1
shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/q_$NAME
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --timed-steps 0 > "$OUT/bench.json" 2> "$OUT/bench.err"
{
  python3 - "$OUT" <<'PY'
import csv, glob, json, os, sys
out = sys.argv[1]
try:
    d = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
    print("steps/s %.1f  us/step %.2f  per_step %s" % (d["value"], 1e3 * d["ms_per_step"], d["per_step"]))
except Exception as e:
    print("bench failed:", e, open(os.path.join(out, "bench.err")).read()[-2000:])
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        short = "deliver" if "deliver_kernel" in n else "neuron" if "neuron_kernel" in n else "reduce" if "reduce_kernel" in n else n[:30]
        print("  %-28s calls %6s avg %10.1f ns  min %8s max %8s" % (short + n[n.find("<"):n.find(">") + 1] if "deliver" in short else short, r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"]))
PY
} > "$ROOT/gpurun_out/q_$NAME.txt" 2>&1
rm -rf "$OUT/trace"
cat "$ROOT/gpurun_out/q_$NAME.txt"
