"""Condenses the rocprofv3 output of profiles/collect.sh into the small files that get committed:
kernel_stats.csv (the --stats table), pmc_<COUNTER>.csv (per-dispatch counter rows of our three kernels only) and
summary.json (per kernel: calls, average duration, average FETCH_SIZE / WRITE_SIZE per launch in KiB as reported,
plus the gfx950-corrected byte counts, see MI355X_MICROARCH.md "HBM")."""
import csv
import glob
import json
import os
import shutil
import sys

out = sys.argv[1]
KERNELS = ("event_deliver_kernel", "deliver_kernel", "neuron_kernel", "reduce_kernel", "remote_push_kernel")  # (first match wins)


def short(name):
    for k in KERNELS:
        if k in name:
            return k
    return None


import time

summary = {"kernels": {}, "collected_unix": int(time.time())}
stats = sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)
if stats:
    shutil.copy(stats[-1], os.path.join(out, "kernel_stats.csv"))
    with open(stats[-1]) as f:
        for row in csv.DictReader(f):
            k = short(row["Name"])
            if k:
                summary["kernels"].setdefault(k, {}).update(calls=int(row["Calls"]), avg_ns=float(row["AverageNs"]),
                                                            min_ns=float(row["MinNs"]), max_ns=float(row["MaxNs"]),
                                                            percent=float(row["Percentage"]))
# The last 20 launches of every kernel = the launches bench.py's roofline pass brackets with HIP events (--timed-steps 20,
# the last steps of the process): their average is the figure to compare with roofline.avg_launch_ms.  The average over
# ALL launches above also holds the device warm-up steps, where the clocks are still settling.
traces = sorted(glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
if traces:
    per = {}
    with open(traces[-1]) as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            if k:
                per.setdefault(k, []).append((int(row["Start_Timestamp"]), int(row["End_Timestamp"]) - int(row["Start_Timestamp"])))
    for k, v in per.items():
        v.sort()
        tail = [d for _, d in v[-20:]]
        summary["kernels"].setdefault(k, {})["avg_ns_last_20_launches"] = sum(tail) / len(tail)
for counter, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    files = sorted(glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        continue
    rows = []
    with open(files[-1]) as f:
        rd = csv.DictReader(f)
        fields = rd.fieldnames
        for row in rd:
            if short(row["Kernel_Name"]) and row["Counter_Name"] == counter:
                rows.append(row)
    with open(os.path.join(out, "pmc_%s.csv" % counter), "w", newline="") as f:
        wr = csv.DictWriter(f, fieldnames=fields)
        wr.writeheader()
        wr.writerows(rows)
    for k in KERNELS:
        vals = [float(r["Counter_Value"]) for r in rows if short(r["Kernel_Name"]) == k]
        if vals:
            # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
            summary["kernels"].setdefault(k, {})["%s_KiB_avg" % counter] = sum(vals) / len(vals)
            summary["kernels"][k]["%s_launches" % counter] = len(vals)
for k, d in summary["kernels"].items():
    if "FETCH_SIZE_KiB_avg" in d:
        # gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes -> double it (MI355X_MICROARCH.md, HBM section)
        d["fetch_bytes_corrected"] = 2.0 * 1024.0 * d["FETCH_SIZE_KiB_avg"]
    if "WRITE_SIZE_KiB_avg" in d:
        d["write_bytes"] = 1024.0 * d["WRITE_SIZE_KiB_avg"]
    if "fetch_bytes_corrected" in d and "write_bytes" in d:
        d["hbm_bytes_per_launch"] = d["fetch_bytes_corrected"] + d["write_bytes"]
for name in ("bench_under_rocprof.json",):
    p = os.path.join(out, name)
    if os.path.exists(p):
        with open(p) as f:
            txt = f.read().strip()
        try:
            summary["bench_line"] = json.loads(txt.splitlines()[-1])
        except Exception:  # noqa: BLE001
            summary["bench_line"] = None
with open(os.path.join(out, "summary.json"), "w") as f:
    json.dump(summary, f, indent=1)
print(json.dumps({k: {x: v.get(x) for x in ("calls", "avg_ns", "hbm_bytes_per_launch")} for k, v in summary["kernels"].items()}))
