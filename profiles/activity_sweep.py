#!/usr/bin/env python3
"""Activity sweep of config C3: timesteps/s and delivery-launch time against the fraction of
neurons that fire per step, for the streaming kernel alone (SANAFE_EVENT=0), the event kernel alone (SANAFE_EVENT=2)
and the per-step device decision (default).  One network (bench.py's --target-activity recipe: threshold 30000, soft
reset, per-neuron biases = rate x threshold), one chip per delivery mode; the activity levels are set by rewriting the
biases (SpikingChip.set_bias) and chip.reset().

    python3 profiles/activity_sweep.py [--levels 0.005,0.01,0.02,0.05,0.1,0.2,0.34] [--modes decide,events,stream]
                                       [--steps 200] [--out gpurun_out/r04_c3_activity.json]

PMC traffic per level comes from `bench.py --target-activity p` (its inline rocprofv3 passes), not from here."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", default="0.005,0.01,0.02,0.05,0.1,0.2,0.34")
    ap.add_argument("--modes", default="decide,events,stream")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--settle", type=int, default=120, help="untimed steps after every bias change (rates down to 1/12 per step)")
    ap.add_argument("--timed-steps", type=int, default=20)
    ap.add_argument("--cores", type=int, default=1024)
    ap.add_argument("--neurons-per-core", type=int, default=256)
    ap.add_argument("--out-degree", type=int, default=2621)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r04_c3_activity.json"))
    args = ap.parse_args()
    levels = [float(x) for x in args.levels.split(",")]
    import _sanafe_pkg
    S = _sanafe_pkg.load()
    n = args.cores * args.neurons_per_core
    arch, net = bench.build_workload(S, 1, args.cores, args.neurons_per_core, args.out_degree, 0.1, args.seed, 0, "int8", levels[0])
    H = S.chip.hip_lib()
    rows = []
    for mode_spec in args.modes.split(","):
        # "events:WAVES=16;LPB=4" -> mode events with SANAFE_EVENT_WAVES=16 SANAFE_EVENT_LPB=4 (tuning runs)
        mode, _, extra = mode_spec.partition(":")
        for k in [k for k in os.environ if k.startswith("SANAFE_EVENT")]:
            os.environ.pop(k)
        for kv in filter(None, extra.split(";")):
            os.environ["SANAFE_EVENT_" + kv.split("=")[0]] = kv.split("=")[1]
        if mode == "events":
            os.environ["SANAFE_EVENT"] = "2"
        elif mode == "stream":
            os.environ["SANAFE_EVENT"] = "0"
        t0 = time.perf_counter()
        chip = S.SpikingChip(arch, device=0)
        chip.load(net)
        load_s = time.perf_counter() - t0
        info = chip.info()
        dev = chip.device_handle()
        lay = (C.c_uint64 * 11)()
        H.sanafe_hip_layout_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        H.sanafe_hip_layout_bytes(dev, lay, 11)
        lay = [int(x) for x in lay]
        for p in levels:
            chip.set_bias("n", bench.activity_biases(n, p, args.seed))
            chip.reset()
            chip.run(args.settle, "simple")
            chip.synchronize()
            pushed0 = chip.device_layout()["pushed_steps"]
            t0 = time.perf_counter()
            tot = chip.run(args.steps, "simple")
            chip.synchronize()
            dt = time.perf_counter() - t0
            dl = chip.device_layout()
            H.sanafe_hip_set_timing(dev, 1)
            if H.sanafe_hip_step(dev, args.timed_steps, 1, 0) != 0:
                raise RuntimeError(H.sanafe_hip_last_error().decode())
            nm, dm, rm, ln = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
            H.sanafe_hip_read_timing(dev, C.byref(nm), C.byref(dm), C.byref(rm), C.byref(ln))
            H.sanafe_hip_set_timing(dev, 0)
            ev, msgs, fired = tot["spikes"] / args.steps, tot["packets_sent"] / args.steps, tot["neurons_fired"] / args.steps
            evl = dl.get("event_layout")
            row = {"mode": mode_spec, "target_activity": p, "fired_per_step": fired, "activity": fired / n,
                   "synaptic_events_per_step": ev, "messages_per_step": msgs,
                   "timesteps_per_s": args.steps / dt, "ms_per_step": 1e3 * dt / args.steps,
                   "steps_delivered_by_events": (dl["pushed_steps"] - pushed0) / float(args.steps),
                   "delivery_launch_ms": dm.value, "neuron_launch_ms": nm.value,
                   "stream_layout_bytes": float(sum(lay[0:5])) + 17.0 * n,
                   "survey_8d_model_bytes": 28.0 * ev + 96.0 * msgs,
                   "event_path_bytes": None if not evl else
                   evl["groups"] * lay[4] + 8.0 * fired * evl["groups"] + ev * lay[9] / max(1.0, float(info["n_synapses"])) +
                   4.0 * evl["segments"] * float(info["n_slots"]),
                   "event_layout": evl, "load_s": load_s}
            rows.append(row)
            sys.stderr.write(json.dumps(row) + "\n")
            sys.stderr.flush()
        del chip
    out = {"workload": "C3 1,024 x 256, out-degree 2,621, --target-activity recipe", "steps": args.steps, "rows": rows}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({"rows": len(rows), "out": args.out}))


if __name__ == "__main__":
    main()
