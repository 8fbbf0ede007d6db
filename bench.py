#!/usr/bin/env python3
"""bench.py -- throughput of the SANA-FE timestep loop on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N=1 by default; N>1 launches its own N ranks)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N>1, one rank per GPU)

Workload (BASELINE.json configs[2], made concrete in SURVEY 8d): arch/loihi_large.yaml + a synthetic
random SNN of 262,144 LIF neurons per GPU (256 neurons on each of 1,024 cores), out-degree 2,621
(1 % of 262,144), integer weights, `loihi_dendrites_delay` dendrites, 10 % of the neurons biased to
fire every step; `simple` timing model on the device.  With N GPUs the tiles are sharded in
contiguous blocks, every GPU holds 262,144 neurons (weak scaling) and the only exchange per step is
an RCCL all-gather of the spike bitmaps.

One "step" = one simulated timestep of the whole chip.  Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


ACTIVITY_THRESHOLD = 30000  # --target-activity: far above anything the +-8 synaptic weights add up to within a run


def activity_biases(n, target, seed):
    """--target-activity: per-neuron integer biases that make `target` of the neurons fire per step in steady state.
    Soft reset (V -= threshold) and a threshold the synaptic input never reaches on its own: a neuron with bias b fires at
    the rate b / threshold.  A fraction q of the neurons gets rates uniform in [1/12, 1/4] (targets up to 12.5 %) or in
    [1/4, 1] (above), q = target / mean rate; the others stay silent.  Different rates keep the population desynchronised,
    and the synaptic input jitters the spike times, so the set of firing neurons changes from step to step."""
    rng = np.random.default_rng(seed + 77)
    lo, hi = (1.0 / 12.0, 0.25) if target <= 0.125 else (0.25, 1.0)
    q = min(1.0, target / (0.5 * (lo + hi)))
    rate = np.where(rng.random(n) < q, lo + (hi - lo) * rng.random(n), 0.0)
    return np.floor(rate * ACTIVITY_THRESHOLD)


def build_workload(S, n_gpus, cores_per_gpu, neurons_per_core, out_degree, p_fire, seed, rank=0, weights="int8", target_activity=None):
    """Built through the product front-end (C++17 / PyBind11 description objects)."""
    n_tiles = max(1024, (cores_per_gpu * n_gpus + 3) // 4) if cores_per_gpu * n_gpus > 4096 else 1024
    arch = S.presets.loihi_large(n_tiles=n_tiles, n_inputs=4)
    cores = arch.cores()
    tiles_per_rank = (n_tiles + n_gpus - 1) // n_gpus
    n_per_gpu = cores_per_gpu * neurons_per_core
    n = n_per_gpu * n_gpus
    net = S.Network("random_%dk" % (n // 1024))
    if target_activity is None:
        attrs = {"threshold": 64, "reset": 0, "force_update": True}
    else:
        attrs = {"threshold": ACTIVITY_THRESHOLD, "reset_mode": "soft", "force_update": True}
    g = net.create_neuron_group("n", n, attrs, "loihi_sparse_synapse", "loihi_dendrites_delay", False, True, "loihi_lif")
    rng = np.random.default_rng(seed)
    if target_activity is None:
        g.set_attribute_column("bias", np.where(rng.random(n) < p_fire, 128.0, 0.0), integer=True)
    else:
        g.set_attribute_column("bias", activity_biases(n, target_activity, seed), integer=True)
    # a rank only needs the edges that start or end in its own shard of the neurons.  Weak scaling: every neuron draws
    # its targets from the n_per_gpu neurons (one GPU's worth of cores) centred on itself, ids wrapping -- on one GPU
    # that is the uniform recipe of SURVEY 8(d); on N GPUs the fan-in statistics of a core (sources per core, synapses
    # per axon) stay what they are on one, and a quarter of the edges cross a GPU boundary
    shard = None if n_gpus == 1 else (rank * n_per_gpu, (rank + 1) * n_per_gpu)
    src, dst, w = S.chip.generate_random_edges(n, out_degree, seed, shard=shard, window=n_per_gpu)
    if weights == "int12":    # integers beyond int8: the 12-bit-weight synapse word (format 3)
        w *= 100.0
    elif weights == "float":  # 16 distinct non-integers: dictionary-coded 2-byte words, fp64 accumulators (format 6)
        w *= 0.7310585786300049
    elif weights in ("int8wide", "floatwide"):
        # ~240 distinct integers in +-127 (4-byte int8 words, format 0) / as many distinct non-integers as synapses
        # (4-byte words + fp64 weights, format 4): the layouts of networks whose weights fit no 32-entry dictionary
        for a in range(0, len(w), 1 << 26):
            b = min(len(w), a + (1 << 26))
            jitter = ((np.arange(a, b, dtype=np.uint64) * np.uint64(2654435761)) >> np.uint64(13)) % np.uint64(15)
            w[a:b] = w[a:b] * 15.0 + (jitter.astype(np.float64) - 7.0)
            if weights == "floatwide":
                w[a:b] *= 0.05 + 1e-9 * jitter
            del jitter
    net.add_edges(src, dst, w, "loihi_sparse_synapse")
    del src, dst, w
    for r in range(n_gpus):
        first_core = r * tiles_per_rank * 4
        for c in range(cores_per_gpu):
            lo = (r * cores_per_gpu + c) * neurons_per_core
            g.map_to_core(cores[first_core + c], lo, lo + neurons_per_core)
    return arch, net


def build_c2(S):
    """BASELINE configs[1]: arch/loihi.yaml + snn/dvs.yaml (18 678 neurons, 3.56 M synapses), rebuilt from the
    committed fixture tests/golden/dvs_yaml.npz (the reference tree does not travel to the GPU box)."""
    import nets
    arch, net = nets.dvs_yaml(S)
    return arch, net


def build_c4(S, n_gpus, rank, tiles_per_gpu, seed):
    """BASELINE configs[3]: arch/truenorth.yaml + synthetic TrueNorth SNN (recipe after
    scripts/tcad2025/compare_nemo_perf.py:52-101): 256 `truenorth` neurons per core, threshold 0,
    reset -1, leak 0, force_update, weight 1, one out-edge per neuron, 80 % of them to another core."""
    n_tiles = tiles_per_gpu * n_gpus
    w = 64 if n_tiles % 64 == 0 else int(np.ceil(np.sqrt(n_tiles)))
    arch = S.presets.truenorth(n_tiles=n_tiles, width=w, height=(n_tiles + w - 1) // w)
    cores = arch.cores()
    npc = 256
    n = n_tiles * npc
    net = S.Network("tn")
    g = net.create_neuron_group("tn", n, {"threshold": 0, "reset": -1, "leak": 0, "force_update": True},
                                "core_synapses", "core_dendrites", False, True, "core_soma")
    rng = np.random.default_rng(seed)
    src = np.arange(n, dtype=np.int64)
    core_of = src // npc
    remote = rng.random(n) < 0.8
    dst_core = np.where(remote, (core_of + 1 + rng.integers(0, max(1, n_tiles - 1), size=n)) % n_tiles, core_of)
    dst = dst_core * npc + rng.integers(0, npc, size=n)
    if n_gpus > 1:  # a rank only needs the edges that start or end in its own tiles
        lo, hi = rank * tiles_per_gpu * npc, (rank + 1) * tiles_per_gpu * npc
        keep = ((src >= lo) & (src < hi)) | ((dst >= lo) & (dst < hi))
        src, dst = src[keep], dst[keep]
    net.add_edges(src, dst, np.ones(len(src)), "core_synapses")
    for c in range(n_tiles):
        g.map_to_core(cores[c], c * npc, (c + 1) * npc)
    return arch, net


def host_info():
    """CPU model, physical cores, logical CPUs and RAM of the machine the baseline is timed on."""
    model, phys = "unknown", set()
    try:
        pid = cid = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name":
                model = v
            elif k == "physical id":
                pid = v
            elif k == "core id":
                cid = v
            elif not k and pid is not None:
                phys.add((pid, cid))
                pid = cid = None
    except OSError:
        pass
    ram = 0
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemTotal"):
                ram = int(line.split()[1]) * 1024
    except OSError:
        pass
    return {"cpu_model": model, "physical_cores": len(phys) or None, "logical_cpus": os.cpu_count(),
            "usable_cpus": len(os.sched_getaffinity(0)), "ram_bytes": ram}


def _time_oracle(chip, timing, seconds, repeats=3):
    """`repeats` timed windows of seconds/repeats each; returns the window with the MEDIAN step rate."""
    for _ in range(2):
        chip.step(timing)
    runs = []
    for _ in range(repeats):
        steps, events, updates, msgs, t0 = 0, 0, 0, 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds / repeats:
            r = chip.step(timing)
            steps += 1
            events += r["spike_count"]
            updates += r["neurons_updated"]
            msgs += r["packets_sent"]
        dt = time.perf_counter() - t0
        runs.append(dict(steps=steps, seconds=dt, events_per_s=events / dt, updates_per_s=updates / dt, steps_per_s=steps / dt,
                         messages_per_s=msgs / dt))
    runs.sort(key=lambda r: r["steps_per_s"])
    med = runs[len(runs) // 2]
    med["all_steps_per_s"] = [r["steps_per_s"] for r in runs]
    return med


def cpu_baseline_same_net(S, args, arch, net, what):
    """The oracle on the very network the GPU ran (small configurations), for about --cpu-seconds: single-threaded and
    with OpenMP over cores on the host's physical cores, like the reference (src/chip.cpp:629-632, 675-678)."""
    from oracle.oracle import OracleChip
    lower = S.cpp.to_desc if isinstance(net, S.cpp.Network) else S.to_desc
    chip = OracleChip(lower(arch, net))
    timing = args.timing
    host = host_info()
    n_all = max(1, min(host["physical_cores"] or host["usable_cpus"], host["usable_cpus"]))
    runs = {}
    for threads in sorted({1, n_all}):
        chip.set_threads(threads)
        runs[threads] = _time_oracle(chip, timing, args.cpu_seconds / 2)
    best_t = max(runs, key=lambda t: runs[t]["steps_per_s"])
    m = runs[best_t]
    return {"value": m["steps_per_s"], "unit": "timesteps/s", "cores": best_t, "kind": "port",
            "sample": "oracle (C++ port of the reference loop, OpenMP over cores), %s timing, on %s: median of 3 windows per thread "
                      "count; value = the faster one (%d threads: %d steps in %.1f s)" % (timing, what, best_t, m["steps"], m["seconds"]),
            "host": host,
            "by_threads": {str(t): {"value": r["steps_per_s"], "synaptic_events_per_s": r["events_per_s"],
                                    "neuron_updates_per_s": r["updates_per_s"], "timesteps_per_s_windows": r["all_steps_per_s"]}
                           for t, r in runs.items()}}


def cpu_baseline(S, args):
    """The oracle (a CPU port of the reference loop) on a bounded sample of the same recipe, single-threaded and with
    the reference's OpenMP-over-cores parallelism (src/chip.cpp:629-632, 675-678) on the host's physical cores."""
    import nets  # noqa: F401
    from oracle.oracle import OracleChip
    cores, npc = 128, 256
    n = cores * npc
    deg = max(8, n // 100)
    arch = S.presets.loihi_large(n_tiles=32, n_inputs=4, width=8, height=4)
    net = S.Network("sample")
    attrs = {"threshold": 64, "reset": 0, "force_update": True} if args.target_activity is None else \
        {"threshold": ACTIVITY_THRESHOLD, "reset_mode": "soft", "force_update": True}
    g = net.create_neuron_group("n", n, attrs, "loihi_sparse_synapse", "loihi_dendrites_delay", False, True, "loihi_lif")
    rng = np.random.default_rng(args.seed)
    if args.target_activity is None:
        g.set_attribute_column("bias", np.where(rng.random(n) < args.p_fire, 128.0, 0.0), integer=True)
    else:
        g.set_attribute_column("bias", activity_biases(n, args.target_activity, args.seed), integer=True)
    src, dst, w = S.chip.generate_random_edges(n, deg, args.seed)
    net.add_edges(src, dst, w, "loihi_sparse_synapse")
    ac = arch.cores()
    for c in range(cores):
        g.map_to_core(ac[c], c * npc, (c + 1) * npc)
    t0 = time.perf_counter()
    chip = OracleChip(S.cpp.to_desc(arch, net))
    build_s = time.perf_counter() - t0
    host = host_info()
    n_all = max(1, min(cores, host["physical_cores"] or host["usable_cpus"], host["usable_cpus"]))
    runs = {}
    for threads in sorted({1, n_all}):
        chip.set_threads(threads)
        # a third of the budget for the single-threaded windows, the rest for all cores (their steps are short)
        m = _time_oracle(chip, args.timing, args.cpu_seconds * (0.6 if threads == 1 else 0.4))
        m["threads"] = threads
        runs[threads] = m
    best = max(runs.values(), key=lambda r: r["events_per_s"])
    best = dict(best)
    best["runs"] = runs
    best["build_s"] = build_s
    best["sample"] = ("oracle (C++ port of the reference loop, OpenMP over cores like src/chip.cpp:629-632, 675-678), %s timing "
                      "(the whole step: neuron + message processing and, under detailed timing, the NoC schedule of "
                      "src/schedule.cpp:208-620) on %d LIF neurons / %d cores, out-degree %d (%d synapses, built in %.1f s): median "
                      "of 3 windows per thread count" % (args.timing, n, cores, deg, n * deg, build_s))
    return best


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=("c3", "c2", "c4"), default="c3",
                    help="c3 (default, the bench line of record): loihi_large + 256k-neuron random SNN per GPU; "
                         "c2: loihi + dvs.yaml (1 GPU); c4: truenorth + 256 neurons x --tiles-per-gpu tiles per GPU")
    ap.add_argument("--tiles-per-gpu", type=int, default=4096, help="c4: TrueNorth tiles (= cores) per GPU")
    ap.add_argument("--timing", choices=("simple", "detailed"), default="simple",
                    help="detailed: NoC schedule on host scheduler threads (1 GPU; reported, not the line of record)")
    ap.add_argument("--scheduler-threads", type=int, default=8)
    ap.add_argument("--cores-per-gpu", type=int, default=1024, help="c3: SURVEY 8(d): 256 neurons on each of 1,024 cores")
    ap.add_argument("--neurons-per-core", type=int, default=256)
    ap.add_argument("--out-degree", type=int, default=2621)
    ap.add_argument("--p-fire", type=float, default=0.1)
    ap.add_argument("--target-activity", type=float, default=None,
                    help="c3: hold the fraction of neurons that fire per step at this value (threshold 30000, soft reset, per-neuron "
                         "biases = rate x threshold) instead of the headline recipe, whose activity runs away to ~34 %%: the regime "
                         "real SNNs run in is 0.5-5 %% (profiles/r04_c3_activity.json)")
    ap.add_argument("--weights", choices=("int8", "int12", "float", "int8wide", "floatwide"), default="int8",
                    help="c3: synaptic weights -- integers in +-8 (default, SURVEY 8d), integers in +-800, 16 non-integers, "
                         "~240 distinct integers in +-127, or a different non-integer per synapse")
    ap.add_argument("--device-warmup", type=int, default=64,
                    help="c3: steps simulated and then undone by chip.reset() before the W warm-up steps -- the GPU's clocks "
                         "need ~30 ms of load to settle (profiles/r02_step_profile.txt); 0 switches it off.  The line carries "
                         "the rate without it as well (value_without_device_warmup)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=16.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-steps", type=int, default=20, help="steps of the event-timed roofline pass")
    ap.add_argument("--traffic", choices=("inline", "stored", "none"), default="inline",
                    help="roofline.traffic (HBM bytes from the PMC counters): inline = two rocprofv3 --pmc child runs of this very "
                         "command (FETCH_SIZE, WRITE_SIZE: separate passes) BEFORE this process touches the GPU; stored = the "
                         "newest profiles/*/summary.json of the same workload (marked as not from this run); none = null")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)  # the run rocprofv3 --pmc wraps: steps only
    ap.add_argument("--exchange", choices=("nccl", "host"), default="nccl",
                    help="N>1 spike exchange: RCCL all-gather on device buffers (default) or gloo through host memory "
                         "(functional check of the multi-rank path on a single GPU)")
    ap.add_argument("--same-device", action="store_true", help="all ranks use device 0 (with --exchange host)")
    ap.add_argument("--force-dist", action="store_true", help="use the N>1 code path (process group + exchange) even with one rank")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as CHILD processes (this process has
    not touched the GPU and never will), relay rank 0's JSON line, fail if any rank fails."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0 = procs[0].stdout.read()
    codes = [p.wait() for p in procs]
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write("bench.py: ranks failed (rank, exit code): %s\n" % bad)
        raise SystemExit(1)
    raise SystemExit(0)


def collect_traffic_inline(args, n_launch_tail):
    """HBM bytes per launch of the two kernels from the PMC counters of THIS command: two child runs under
    `rocprofv3 --kernel-trace --pmc <counter>` (FETCH_SIZE and WRITE_SIZE do not share a pass; --pmc is never combined
    with API tracing), each simulating the same steps as this process; averaged over the last `n_launch_tail` launches
    (the steps the event-timed roofline pass brackets).  gfx950 corrections as MI355X_MICROARCH.md prescribes:
    FETCH_SIZE in KiB, doubled (128-byte requests tallied at 64); WRITE_SIZE in KiB.  Runs BEFORE this process touches
    the GPU (children only).  Returns None on any failure (the line then says so)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not on PATH"
    total_steps = args.warmup + args.steps + args.timed_steps
    passthrough = []
    skip = 0
    for a in sys.argv[1:]:
        if skip:
            skip -= 1
            continue
        if a in ("--steps", "--warmup", "--device-warmup", "--timed-steps", "--traffic", "--cpu-seconds"):
            skip = 1
            continue
        if a.startswith(("--steps=", "--warmup=", "--device-warmup=", "--timed-steps=", "--traffic=", "--cpu-seconds=")) \
                or a == "--no-cpu-baseline":
            continue
        passthrough.append(a)
    res = {}
    tmp = tempfile.mkdtemp(prefix="sanafe_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.abspath(__file__)] + passthrough + \
                  ["--pmc-child", "--steps", str(total_steps), "--warmup", "0", "--device-warmup", "0", "--timed-steps", "0",
                   "--no-cpu-baseline", "--traffic", "none"]
            try:
                p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL,
                                   stderr=subprocess.PIPE, timeout=420)
            except subprocess.TimeoutExpired:
                return None, "rocprofv3 --pmc %s child timed out" % counter
            if p.returncode != 0:
                return None, "rocprofv3 --pmc %s child failed (%d): %s" % (counter, p.returncode, p.stderr.decode()[-300:])
            rows = {}
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] != counter:
                            continue
                        k = "event_deliver_kernel" if "event_deliver_kernel" in row["Kernel_Name"] else \
                            "deliver_kernel" if "deliver_kernel" in row["Kernel_Name"] else \
                            "neuron_kernel" if "neuron_kernel" in row["Kernel_Name"] else None
                        if k:
                            rows.setdefault(k, []).append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
            for k, v in rows.items():
                v.sort()
                tail = [x for _, x in v[-n_launch_tail:]]
                res.setdefault(k, {})[counter + "_KiB_avg"] = sum(tail) / len(tail)
                res[k][counter + "_launches_averaged"] = len(tail)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    for k, d in res.items():
        if "FETCH_SIZE_KiB_avg" in d and "WRITE_SIZE_KiB_avg" in d:
            d["fetch_bytes_corrected"] = 2.0 * 1024.0 * d["FETCH_SIZE_KiB_avg"]
            d["write_bytes"] = 1024.0 * d["WRITE_SIZE_KiB_avg"]
            d["hbm_bytes_per_launch"] = d["fetch_bytes_corrected"] + d["write_bytes"]
    if not any("hbm_bytes_per_launch" in d for d in res.values()):
        return None, "no neuron_kernel / deliver_kernel counter rows in the rocprofv3 output"
    return res, None


def main():
    args = parse_args()
    if "RANK" not in os.environ and args.gpus > 1 and not args.pmc_child:
        launch_ranks(args)  # does not return

    # Libraries (RCCL's version banner, gloo's connection notes) write to stdout; the contract is ONE
    # JSON line there, so everything else is sent to stderr and the line goes to the saved descriptor.
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    os.environ.setdefault("NCCL_DEBUG", "WARN")

    if "RANK" not in os.environ:  # plain `python bench.py` (N=1): a self-contained rendezvous for --force-dist
        os.environ.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    multi = world > 1 or args.force_dist
    want_roof = args.timed_steps > 0 and args.timing == "simple" and not multi and not args.pmc_child
    # ---- PMC passes first: children only, before this process initialises the GPU ----
    pmc, pmc_error = None, None
    if want_roof and args.traffic == "inline":
        t_pmc = time.perf_counter()
        pmc, pmc_error = collect_traffic_inline(args, args.timed_steps)
        t_pmc = time.perf_counter() - t_pmc
    import _sanafe_pkg
    S = _sanafe_pkg.load()

    t_setup = time.perf_counter()
    if args.workload == "c2":
        if world > 1:
            raise SystemExit("workload c2 is a single-GPU configuration")
        arch, net = build_c2(S)
        n_total = 18678
        workload = "arch/loihi.yaml + snn/dvs.yaml (BASELINE configs[1]): 18678 LIF neurons on 49 cores, 3564441 synapses"
    elif args.workload == "c4":
        arch, net = build_c4(S, world, rank, args.tiles_per_gpu, args.seed)
        n_total = args.tiles_per_gpu * 256 * world
        workload = ("arch/truenorth.yaml + synthetic TrueNorth SNN (BASELINE configs[3]): %d neurons (%d tiles x 256 per GPU), "
                    "one out-edge per neuron, 80%% remote" % (n_total, args.tiles_per_gpu))
    else:
        arch, net = build_workload(S, world, args.cores_per_gpu, args.neurons_per_core, args.out_degree, args.p_fire,
                                   args.seed, rank, args.weights, args.target_activity)
        n_total = args.cores_per_gpu * args.neurons_per_core * world
        workload = ("arch/loihi_large.yaml + synthetic random SNN (BASELINE configs[2]): %d LIF neurons (%d cores x %d per "
                    "GPU), out-degree %d, %s, loihi_dendrites_delay"
                    % (n_total, args.cores_per_gpu, args.neurons_per_core, args.out_degree,
                       "%.0f%% biased to fire every step" % (100 * args.p_fire) if args.target_activity is None else
                       "activity held at %.3g%% of the neurons per step (threshold %d, soft reset, per-neuron biases)"
                       % (100 * args.target_activity, ACTIVITY_THRESHOLD)))
        if args.weights != "int8":
            workload += ", %s weights" % args.weights
    workload += ", %s timing" % args.timing
    t_net = time.perf_counter() - t_setup
    chip = S.SpikingChip(arch, device=0 if args.same_device else local_rank, n_ranks=world, rank=rank)
    chip.load(net)
    info = chip.info()
    t_load = time.perf_counter() - t_setup - t_net
    H = S.chip.hip_lib()
    dev = chip.device_handle()

    dist = None
    if multi:
        # Control plane (rendezvous, barrier, max-over-ranks of the wall time): torch.distributed over gloo.
        # Data plane: the product's own per-step spike exchange inside chip.sim() -- RCCL all-gather on the device
        # bitmap (host/comm.cpp), or the host callback path for two ranks sharing one GPU (--exchange host).
        import torch.distributed as dist_mod
        dist = dist_mod
        dist.init_process_group("gloo")
        chip.comm_init_torch(dist, "rccl" if args.exchange == "nccl" else "host")
    if args.timing == "detailed":
        L = S.chip.lib()
        L.sanafe_chip_set_scheduler_threads(chip._h, args.scheduler_threads)

    def run_steps(k):
        # one SpikingChip.sim-level call: on N ranks it returns the totals of the WHOLE chip on every rank
        return chip.run(k, args.timing)

    def sync():
        chip.synchronize()

    def timed_region():
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; max over ranks."""
        run_steps(args.warmup)
        sync()
        if dist:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        data = run_steps(args.steps)
        sync()
        if dist:
            dist.barrier()
        dt = time.perf_counter() - t0
        if dist:
            import torch
            tmax = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax[0])
        return data, dt

    if args.pmc_child:  # the run rocprofv3 --pmc wraps: the same steps of the same simulation, nothing else
        run_steps(args.steps)
        sync()
        return

    # Device warm-up, NOT part of the workload: the first ~30 ms of load after an idle period run at unsettled clocks
    # (the same steps repeated after a reset are 10-25 % faster, profiles/r02_step_profile.txt).  The line carries BOTH
    # rates: first steps 1..W+K right after load() (value_without_device_warmup), then -- after chip.reset(), the device
    # warm-up steps and another chip.reset() -- steps 1..W+K of the same simulation again at settled clocks (value).
    device_warmup = args.device_warmup if args.workload == "c3" and args.timing == "simple" else 0
    cold = None
    if device_warmup > 0:
        _, cold_elapsed = timed_region()
        cold = args.steps / cold_elapsed
        chip.reset()
        run_steps(device_warmup)
        sync()
        chip.reset()
    run_data, elapsed = timed_region()
    agg = {k: float(run_data[k]) for k in ("spikes", "packets_sent", "neurons_updated", "neurons_fired")}

    # ---- roofline pass: HIP events around every kernel on its own stream (one rank, the chip's own stream) ----
    roof = None
    if want_roof:
        pushed_before = chip.device_layout()["pushed_steps"]
        H.sanafe_hip_set_timing(dev, 1)
        b2 = chip.read_totals()
        if H.sanafe_hip_step(dev, args.timed_steps, 1, 0) != 0:
            raise RuntimeError(H.sanafe_hip_last_error().decode())
        a2 = chip.read_totals()
        nm, dm, rm, ln = C.c_double(), C.c_double(), C.c_double(), C.c_int64()
        H.sanafe_hip_read_timing(dev, C.byref(nm), C.byref(dm), C.byref(rm), C.byref(ln))
        H.sanafe_hip_set_timing(dev, 0)
        ev = (a2["spikes"] - b2["spikes"]) / args.timed_steps
        msgs = (a2["packets_sent"] - b2["packets_sent"]) / args.timed_steps
        upd = (a2["neurons_updated"] - b2["neurons_updated"]) / args.timed_steps
        fired = (a2["neurons_fired"] - b2["neurons_fired"]) / args.timed_steps
        # (1) ALGORITHMIC bytes of one launch = what THIS layout has to move, from the chip's own tables
        #     (sanafe_hip_layout_bytes): delivery = synapse words + axon records + chunk tables + slice descriptors + spike
        #     bitmap, each read once when every chunk is streamed (an upper bound when few axons spike), + one 17-byte
        #     write-back per neuron; neuron launch = per-slot state read + written, + 40 B per fired neuron.  DESIGN.md 5
        #     states the per-unit figures.  achieved = these bytes / the HIP-event-timed launch duration of THIS run.
        lay = (C.c_uint64 * 11)()
        H.sanafe_hip_layout_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        if H.sanafe_hip_layout_bytes(dev, lay, 11) != 0:
            raise RuntimeError(H.sanafe_hip_last_error().decode())
        lay = [int(x) for x in lay]
        design_stream = float(sum(lay[0:5])) + 17.0 * upd
        design_deliver, deliver_parts, deliver_name = design_stream, None, "deliver_kernel"
        layout_now = chip.device_layout()
        evl = layout_now.get("event_layout")
        by_events = (layout_now["pushed_steps"] - pushed_before) / float(args.timed_steps) if evl else 0.0
        if evl:
            # Steps delivered by events (event_deliver_kernel) read: every workgroup its segment of the spike bitmap
            # (groups x bitmap), per fired neuron and core group one 8-byte entry of the block table, the fired neurons'
            # blocks of 2-byte words (padded to 16 bytes per block: layout bytes / synapses per word on average), and write
            # one 4-byte partial per (segment, neuron slot) -- every accumulator, touched or not -- which the next neuron
            # launch reads back (counted there)
            ev_parts = {"spike_bitmap_scans": evl["groups"] * lay[4], "block_tables": 8.0 * fired * evl["groups"],
                        "synapse_words_of_fired_neurons": ev * lay[9] / max(1.0, float(info["n_synapses"])),
                        "partial_rows_written": 4.0 * evl["segments"] * float(info["n_slots"])}
            design_event = float(sum(ev_parts.values()))
            design_deliver = by_events * design_event + (1.0 - by_events) * design_stream
            if by_events >= 0.5:
                deliver_name, deliver_parts = "event_deliver_kernel", ev_parts
        design_neuron = float(lay[5] + lay[6]) + lay[7] * fired
        if evl:  # after a step delivered by events the neuron launch reads all eight partial rows instead of the buffer row
            design_neuron += by_events * 4.0 * 8.0 * float(info["n_slots"])
        # (2) SURVEY 8(d)'s byte model (48 B per neuron update, 28 B per synaptic event, 96 B per message): it prices
        #     12-byte synapses, HBM accumulators and materialised 40-byte messages, none of which this design moves, so
        #     on dense workloads it exceeds 1 -- kept as a secondary figure only (frac_contract_model).
        contract_deliver = 28.0 * ev + 96.0 * msgs

        def gbps(nbytes, ms):
            return nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0

        achieved = gbps(design_deliver, dm.value)
        roof = {"bound": "hbm", "kernel": deliver_name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None, "traffic_from_this_run": False,
                "algorithmic_bytes_per_launch": design_deliver, "avg_launch_ms": dm.value,
                "steps_delivered_by_events": by_events,
                "algorithmic_bytes_parts": deliver_parts or {"synapse_words": lay[0], "axon_records": lay[1], "chunk_tables": lay[2],
                                                             "slice_descriptors": lay[3], "spike_bitmap": lay[4],
                                                             "write_back": 17.0 * upd},
                # bitmap axon records keep one synapse-count byte per axon that only the gather path reads (windows with
                # fewer than 8 spiking axons): NOT part of algorithmic_bytes_per_launch
                "axon_record_bytes_read_by_gather_path_only": lay[8],
                "units_per_launch": {"synaptic_events": ev, "messages": msgs, "neurons_updated": upd, "neurons_fired": fired,
                                     "synapses_streamed": int(info["n_synapses"]), "axons_probed": int(info["n_axons"])},
                "contract_model_bytes_per_launch": contract_deliver,
                "frac_contract_model": gbps(contract_deliver, dm.value) / HBM_PEAK_GBS,
                "neuron_kernel": {"avg_launch_ms": nm.value, "algorithmic_bytes_per_launch": design_neuron,
                                  "achieved_GBps": gbps(design_neuron, nm.value),
                                  "frac": gbps(design_neuron, nm.value) / HBM_PEAK_GBS,
                                  "contract_model_bytes_per_launch": 48.0 * upd,
                                  "frac_contract_model": gbps(48.0 * upd, nm.value) / HBM_PEAK_GBS},
                "reduce_kernel_avg_ms": rm.value, "launches": ln.value,
                "whole_step": {"algorithmic_bytes": design_neuron + design_deliver,
                               "achieved_GBps": gbps(design_neuron + design_deliver, nm.value + dm.value + rm.value)}}

    if roof is not None and roof["neuron_kernel"]["avg_launch_ms"] > roof["avg_launch_ms"]:
        # The neuron launch dominates the step (C4: a push-only chip has no delivery launch at all; C2): the top-level
        # fields describe THAT kernel, the delivery launch moves to a sub-block.
        nk = roof.pop("neuron_kernel")
        roof["deliver_kernel"] = {k: roof[k] for k in ("avg_launch_ms", "algorithmic_bytes_per_launch", "achieved", "frac",
                                                       "contract_model_bytes_per_launch", "frac_contract_model",
                                                       "algorithmic_bytes_parts")}
        roof.update({"kernel": "neuron_kernel", "achieved": nk["achieved_GBps"], "frac": nk["frac"],
                     "algorithmic_bytes_per_launch": nk["algorithmic_bytes_per_launch"], "avg_launch_ms": nk["avg_launch_ms"],
                     "contract_model_bytes_per_launch": nk["contract_model_bytes_per_launch"],
                     "frac_contract_model": nk["frac_contract_model"],
                     "algorithmic_bytes_parts": {"per_slot_state_read": lay[5], "per_slot_state_written": lay[6],
                                                 "per_fired_neuron": lay[7] * fired}})
    dominant = roof["kernel"] if roof is not None else None
    if roof is not None:
        # (3) HBM bytes per launch from the PMC counters (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950,
        #     plus WRITE_SIZE; separate --pmc passes): collected by this very command in two rocprofv3 child runs
        #     (--traffic inline, the default), or -- marked as NOT from this run -- read from the newest stored
        #     profiles/*/summary.json of the same workload.
        if pmc is not None and "hbm_bytes_per_launch" in pmc.get(dominant, {}):
            roof["traffic"] = pmc[dominant]["hbm_bytes_per_launch"]
            roof["traffic_from_this_run"] = True
            roof["traffic_source"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE child runs of this command "
                                      "(%.0f s), averaged over the last %d launches" % (t_pmc, args.timed_steps))
            roof["traffic_counters"] = pmc[dominant]
            if "neuron_kernel" in roof and "hbm_bytes_per_launch" in pmc.get("neuron_kernel", {}):
                roof["neuron_kernel"]["traffic"] = pmc["neuron_kernel"]["hbm_bytes_per_launch"]
        elif args.traffic != "none":
            if pmc_error:
                roof["traffic_inline_error"] = pmc_error
            import glob
            newest = -1
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "summary.json"))):
                try:
                    with open(path) as f:
                        summ = json.load(f)
                    if summ["bench_line"]["config"]["workload"] == workload and "hbm_bytes_per_launch" in summ["kernels"][dominant] \
                            and summ.get("collected_unix", 0) > newest:
                        newest = summ.get("collected_unix", 0)
                        roof["traffic"] = summ["kernels"][dominant]["hbm_bytes_per_launch"]
                        roof["traffic_source"] = "STORED, not from this run: " + os.path.relpath(path, ROOT)
                        roof["traffic_collected_utc"] = time.strftime("%Y-%m-%d %H:%M", time.gmtime(newest)) if newest > 0 else None
                        nk = summ["kernels"].get("neuron_kernel", {})
                        if "neuron_kernel" in roof and "hbm_bytes_per_launch" in nk:
                            roof["neuron_kernel"]["traffic"] = nk["hbm_bytes_per_launch"]
                except (KeyError, TypeError, ValueError, OSError):
                    continue
        if roof["traffic"]:
            roof["traffic_GBps"] = roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9
            roof["traffic_frac_of_peak"] = roof["traffic_GBps"] / HBM_PEAK_GBS
            roof["traffic_over_algorithmic"] = roof["traffic"] / roof["algorithmic_bytes_per_launch"]
        roof["note"] = ("achieved = algorithmic_bytes_per_launch (the bytes of the chip's own device layout, "
                        "algorithmic_bytes_parts: %.2f B per synapse word, %.2f B per axon record + tables, 17 B per written-back "
                        "neuron) / avg_launch_ms (HIP events on the kernels' stream, this run); frac = achieved / peak.  "
                        "frac_contract_model prices the same launch with SURVEY 8(d)'s model (28 B per synaptic event + 96 B per "
                        "message: materialised 40-byte messages, 12-byte synapses, HBM accumulators -- %.1fx the bytes this "
                        "design moves) and is not a roofline fraction.  traffic = FETCH_SIZE x 2 + WRITE_SIZE per launch."
                        % (lay[0] / max(1.0, float(info["n_synapses"])), (lay[1] + lay[2]) / max(1.0, float(info["n_axons"])),
                           contract_deliver / max(1.0, design_deliver)))

    cpu = None
    if rank == 0 and not multi and not args.no_cpu_baseline and args.workload == "c3":
        c = cpu_baseline(S, args)
        events_per_step = agg["spikes"] / args.steps

        def scaled(m):
            return m["events_per_s"] / events_per_step if events_per_step > 0 else m["steps_per_s"]
        cpu = {"value": scaled(c), "unit": "timesteps/s", "cores": c["threads"], "kind": "port", "host": host_info(),
               "sample": c["sample"] + "; value = synaptic-events/s of the faster thread count (%d threads: %.3g) / events per "
                         "step of the GPU workload (%.3g)" % (c["threads"], c["events_per_s"], events_per_step),
               "by_threads": {str(t): {"value": scaled(m), "timesteps_per_s_on_sample": m["steps_per_s"],
                                       "synaptic_events_per_s": m["events_per_s"], "neuron_updates_per_s": m["updates_per_s"],
                                       "messages_per_s": m["messages_per_s"],
                                       "timesteps_per_s_windows": m["all_steps_per_s"]} for t, m in c["runs"].items()}}

    if rank == 0 and not multi and not args.no_cpu_baseline and args.workload == "c2":
        cpu = cpu_baseline_same_net(S, args, arch, net, "the same network")
    if rank == 0 and not multi and not args.no_cpu_baseline and args.workload == "c4":
        a2, n2 = build_c4(S, 1, 0, 256, args.seed)
        cpu = cpu_baseline_same_net(S, args, a2, n2, "a 256-tile (65536-neuron) sample of the same recipe")
        cpu["value"] *= 256.0 / (args.tiles_per_gpu * world)  # per-step cost of this recipe is linear in tiles
        cpu["sample"] += "; value scaled by 256 / %d tiles" % (args.tiles_per_gpu * world)
    if rank == 0:
        out = {
            "metric": "simulated timesteps/sec", "value": args.steps / elapsed, "unit": "timesteps/s",
            "value_without_device_warmup": cold,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload,
                       "neurons": n_total, "synapses_per_gpu": int(info["n_synapses"]), "axons_per_gpu": int(info["n_axons"]),
                       "timing_model": args.timing, "device_layout": chip.device_layout(),
                       "exchange": ("in-place RCCL all-gather of the spike bitmap windows inside chip.sim() (host/comm.cpp)"
                                    if args.exchange == "nccl" else "host all-gather callback (gloo)") if multi else "none"},
            "totals_in_timed_region": agg,
            "neuron_updates_per_s": agg["neurons_updated"] / elapsed,
            "synaptic_events_per_s": agg["spikes"] / elapsed, "messages_per_s": agg["packets_sent"] / elapsed,
            "per_step": {"neurons_updated": agg["neurons_updated"] / args.steps, "neurons_fired": agg["neurons_fired"] / args.steps,
                         "synaptic_events": agg["spikes"] / args.steps, "messages": agg["packets_sent"] / args.steps},
            "setup_s": {"build_network": t_net, "map_and_upload": t_load},
            "device_warmup": {"steps": device_warmup, "then": "chip.reset()",
                              "note": "value: clock warm-up steps + chip.reset() before the W warm-up steps, so the timed steps are "
                                      "steps W+1..W+K of a fresh simulation at settled clocks; value_without_device_warmup: the same "
                                      "W+K steps timed first, right after load()"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
