#!/bin/bash
# Round-4 evidence in one gpurun call: the bench lines of record (bench.py's own inline PMC passes) and the
# rocprofv3 --kernel-trace --stats summaries of the same commands (kernel time, not counters: --traffic none inside).
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out
cd "$ROOT"
echo "[r04] c3 line of record (driver's command)"
timeout -k 10 600 python3 bench.py --steps 20 --warmup 5 > $OUT/r04_bench_c3.json 2> $OUT/r04_bench_c3.err
echo "[r04] c3 3000 steps"
timeout -k 10 400 python3 bench.py --steps 3000 --warmup 100 --traffic none --no-cpu-baseline > $OUT/r04_bench_c3_3000steps.json 2> $OUT/r04_bench_c3_3000.err
echo "[r04] c4, c2"
timeout -k 10 300 python3 bench.py --workload c4 --steps 3000 --warmup 100 > $OUT/r04_bench_c4_1gpu.json 2> $OUT/r04_bench_c4.err
timeout -k 10 300 python3 bench.py --workload c2 --steps 3000 --warmup 100 > $OUT/r04_bench_c2_simple.json 2> $OUT/r04_bench_c2.err
cd /tmp; export TMPDIR=/tmp
for spec in "c3:--steps 20 --warmup 5" "c3_act002:--target-activity 0.02 --steps 200 --warmup 150" "c4:--workload c4 --steps 3000 --warmup 100" "c2:--workload c2 --steps 3000 --warmup 100"; do
    name=${spec%%:*}; args=${spec#*:}
    echo "[r04] rocprofv3 --stats $name"
    d=$OUT/prof_r04_$name; rm -rf "$d"; mkdir -p "$d"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$d/trace" -- python3 "$ROOT/bench.py" $args --traffic none --no-cpu-baseline > "$d/bench_under_rocprof.json" 2> "$d/trace.err"
    python3 "$ROOT/profiles/summarize.py" "$d" > /dev/null 2>&1
    rm -rf "$d/trace"
done
python3 - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
for f in ("r04_bench_c3", "r04_bench_c3_3000steps", "r04_bench_c4_1gpu", "r04_bench_c2_simple"):
    try:
        d = json.loads(open(os.path.join(out, f + ".json")).read().strip().splitlines()[-1]); r = d["roofline"]
        print(f, round(d["value"]), "cold", d.get("value_without_device_warmup"), "ms/step", d["ms_per_step"], r["kernel"], r["avg_launch_ms"], "frac", round(r["frac"], 3),
              "traffic", r.get("traffic"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e:
        print(f, "ERR", e)
for n in ("c3", "c3_act002", "c4", "c2"):
    try:
        s = json.load(open(os.path.join(out, "prof_r04_" + n, "summary.json")))
        print(n, {k: (v.get("calls"), round(v.get("avg_ns", 0) / 1e3, 2), round(v.get("avg_ns_last_20_launches", 0) / 1e3, 2)) for k, v in s["kernels"].items()})
    except Exception as e:
        print(n, "ERR", e)
PY
# SQ counters of the event kernel at the headline's activity (two --pmc passes; profiles/pmc_sq.sh)
bash "$ROOT/profiles/pmc_sq.sh" r04_event --steps 100 --warmup 20 > "$OUT/r04_sq_event.txt" 2>&1
cp "$OUT/sq_r04_event/sq_summary.json" "$OUT/r04_sq_event.json" 2>/dev/null
