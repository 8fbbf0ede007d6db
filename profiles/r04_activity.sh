#!/bin/bash
# profiles/r04_c3_activity.json: the activity sweep of C3 1,024 x 256 -- per-step device decision,
# event kernel alone, streaming kernel alone -- plus bench.py lines with PMC traffic at three activity levels.
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOT"
OUT=$ROOT/gpurun_out
timeout -k 10 900 python3 profiles/activity_sweep.py --levels 0.005,0.01,0.02,0.05,0.1,0.2,0.34,0.5 --modes decide,events,stream --out $OUT/r04_sweep.json 2> $OUT/r04_sweep.err
for p in 0.005 0.02 0.1; do
    timeout -k 10 500 python3 bench.py --target-activity $p --steps 200 --warmup 150 --no-cpu-baseline > $OUT/r04_bench_act_$p.json 2> $OUT/r04_bench_act_$p.err
done
python3 - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
d = json.load(open(os.path.join(out, "r04_sweep.json")))
d["bench_lines_with_pmc_traffic"] = {}
for p in ("0.005", "0.02", "0.1"):
    try:
        d["bench_lines_with_pmc_traffic"][p] = json.loads(open(os.path.join(out, "r04_bench_act_%s.json" % p)).read().strip().splitlines()[-1])
    except Exception as e:
        d["bench_lines_with_pmc_traffic"][p] = {"error": str(e)}
json.dump(d, open(os.path.join(out, "r04_c3_activity.json"), "w"), indent=1)
print("%-8s %-7s %-9s %-10s %-11s %-6s" % ("mode", "act", "fired", "steps/s", "deliver_ms", "byEv"))
for r in d["rows"]:
    print("%-8s %-7.3f %-9.0f %-10.0f %-11.4f %-6.2f" % (r["mode"], r["activity"], r["fired_per_step"], r["timesteps_per_s"], r["delivery_launch_ms"], r["steps_delivered_by_events"]))
for p, b in d["bench_lines_with_pmc_traffic"].items():
    r = b.get("roofline") or {}
    print("bench", p, round(b.get("value", 0)), r.get("kernel"), r.get("avg_launch_ms"), "alg", r.get("algorithmic_bytes_per_launch"), "pmc", r.get("traffic"), "frac", r.get("frac"))
PY
