#!/bin/bash
# SQ counters of one bench configuration: profiles/pmc_sq.sh <name> [bench args]
set -eo pipefail
NAME=$1
shift
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/sq_$NAME
mkdir -p "$OUT"
cd /tmp
export TMPDIR=/tmp
rocprofv3 -L > "$OUT/counters.txt" 2>&1 || true
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d "$OUT/p1" -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --timed-steps 0 > "$OUT/b1.json" 2> "$OUT/p1.err"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d "$OUT/p2" -- python3 "$ROOT/bench.py" "$@" --no-cpu-baseline --timed-steps 0 > "$OUT/b2.json" 2> "$OUT/p2.err"
python3 - "$OUT" <<'PY'
import csv, glob, os, sys, json
out = sys.argv[1]
res = {}
for sub in ("p1", "p2"):
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"]
            name = "event_deliver" if "event_deliver_kernel" in k else "deliver" if "deliver_kernel" in k else "neuron" if "neuron_kernel" in k else "reduce" if "reduce_kernel" in k else None
            if not name:
                continue
            d = res.setdefault(name, {}).setdefault(row["Counter_Name"], [0.0, 0])
            d[0] += float(row["Counter_Value"]); d[1] += 1
summary = {k: {c: v[0] / v[1] for c, v in d.items()} for k, d in res.items()}
json.dump(summary, open(os.path.join(out, "sq_summary.json"), "w"), indent=1)
print(json.dumps({k: summary[k] for k in ("event_deliver", "deliver") if k in summary}, indent=1))
PY
