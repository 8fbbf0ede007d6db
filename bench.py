#!/usr/bin/env python3
"""bench.py -- throughput of the SANA-FE timestep loop on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N=1 by default)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (N>1, one rank per GPU)

Workload (BASELINE.json configs[2], made concrete in SURVEY 8d): arch/loihi_large.yaml + a synthetic
random SNN of 262,144 LIF neurons per GPU (512 neurons on each of 512 cores), out-degree 2,621
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


def build_workload(S, n_gpus, cores_per_gpu, neurons_per_core, out_degree, p_fire, seed, rank=0, weights="int8"):
    """Built through the product front-end (C++17 / PyBind11 description objects)."""
    n_tiles = max(1024, (cores_per_gpu * n_gpus + 3) // 4) if cores_per_gpu * n_gpus > 4096 else 1024
    arch = S.presets.loihi_large(n_tiles=n_tiles, n_inputs=4)
    cores = arch.cores()
    tiles_per_rank = (n_tiles + n_gpus - 1) // n_gpus
    n_per_gpu = cores_per_gpu * neurons_per_core
    n = n_per_gpu * n_gpus
    net = S.Network("random_%dk" % (n // 1024))
    g = net.create_neuron_group("n", n, {"threshold": 64, "reset": 0, "force_update": True}, "loihi_sparse_synapse",
                                "loihi_dendrites_delay", False, True, "loihi_lif")
    rng = np.random.default_rng(seed)
    g.set_attribute_column("bias", np.where(rng.random(n) < p_fire, 128.0, 0.0), integer=True)
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
    for _ in range(3):
        chip.step(timing)
    runs = []
    for _ in range(repeats):
        steps, events, updates, t0 = 0, 0, 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds / repeats:
            r = chip.step(timing)
            steps += 1
            events += r["spike_count"]
            updates += r["neurons_updated"]
        dt = time.perf_counter() - t0
        runs.append(dict(steps=steps, seconds=dt, events_per_s=events / dt, updates_per_s=updates / dt, steps_per_s=steps / dt))
    runs.sort(key=lambda r: r["steps_per_s"])
    med = runs[len(runs) // 2]
    med["all_steps_per_s"] = [r["steps_per_s"] for r in runs]
    return med


def cpu_baseline_same_net(S, args, arch, net, what):
    """The oracle on the very network the GPU ran (small configurations), for about --cpu-seconds."""
    from oracle.oracle import OracleChip
    lower = S.cpp.to_desc if isinstance(net, S.cpp.Network) else S.to_desc
    chip = OracleChip(lower(arch, net))
    timing = args.timing
    m = _time_oracle(chip, timing, args.cpu_seconds)
    return {"value": m["steps_per_s"], "unit": "timesteps/s", "cores": 1, "kind": "port",
            "sample": "oracle (scalar C++ port, 1 thread), %s timing, on %s: median of 3 windows (%d steps in %.1f s)"
                      % (timing, what, m["steps"], m["seconds"]),
            "host": host_info(),
            "measured": {"synaptic_events_per_s": m["events_per_s"], "neuron_updates_per_s": m["updates_per_s"],
                         "timesteps_per_s_windows": m["all_steps_per_s"]}}


def cpu_baseline(S, args):
    """The oracle (a scalar CPU port of the reference loop) on a bounded sample of the same recipe."""
    import nets  # noqa: F401
    from oracle.oracle import OracleChip
    cores, npc = 64, 256
    n = cores * npc
    deg = max(8, n // 100)
    arch = S.presets.loihi_large(n_tiles=16, n_inputs=4, width=4, height=4)
    net = S.Network("sample")
    g = net.create_neuron_group("n", n, {"threshold": 64, "reset": 0, "force_update": True}, "loihi_sparse_synapse",
                                "loihi_dendrites_delay", False, True, "loihi_lif")
    rng = np.random.default_rng(args.seed)
    g.set_attribute_column("bias", np.where(rng.random(n) < args.p_fire, 128.0, 0.0), integer=True)
    src, dst, w = S.chip.generate_random_edges(n, deg, args.seed)
    net.add_edges(src, dst, w, "loihi_sparse_synapse")
    ac = arch.cores()
    for c in range(cores):
        g.map_to_core(ac[c], c * npc, (c + 1) * npc)
    chip = OracleChip(S.cpp.to_desc(arch, net))
    m = _time_oracle(chip, "simple", args.cpu_seconds)
    m["sample"] = ("oracle (scalar C++ port, 1 thread) on %d LIF neurons / %d cores, out-degree %d: median of 3 windows "
                   "(%d steps in %.1f s)" % (n, cores, deg, m["steps"], m["seconds"]))
    return m


def main():
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
    ap.add_argument("--cores-per-gpu", type=int, default=512)
    ap.add_argument("--neurons-per-core", type=int, default=512)
    ap.add_argument("--out-degree", type=int, default=2621)
    ap.add_argument("--p-fire", type=float, default=0.1)
    ap.add_argument("--weights", choices=("int8", "int12", "float", "int8wide", "floatwide"), default="int8",
                    help="c3: synaptic weights -- integers in +-8 (default, SURVEY 8d), integers in +-800, 16 non-integers, "
                         "~240 distinct integers in +-127, or a different non-integer per synapse")
    ap.add_argument("--device-warmup", type=int, default=64,
                    help="c3: steps simulated and then undone by chip.reset() before the W warm-up steps -- the GPU's clocks "
                         "need ~30 ms of load to settle (profiles/r02_step_profile.txt); 0 switches it off")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-steps", type=int, default=20, help="steps of the event-timed roofline pass")
    ap.add_argument("--exchange", choices=("nccl", "host"), default="nccl",
                    help="N>1 spike exchange: RCCL all-gather on device buffers (default) or gloo through host memory "
                         "(functional check of the multi-rank path on a single GPU)")
    ap.add_argument("--same-device", action="store_true", help="all ranks use device 0 (with --exchange host)")
    ap.add_argument("--force-dist", action="store_true", help="use the N>1 code path (process group + exchange) even with one rank")
    args = ap.parse_args()

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
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
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
                                   args.seed, rank, args.weights)
        n_total = args.cores_per_gpu * args.neurons_per_core * world
        workload = ("arch/loihi_large.yaml + synthetic random SNN (BASELINE configs[2]): %d LIF neurons (%d cores x %d per "
                    "GPU), out-degree %d, %.0f%% biased to fire every step, loihi_dendrites_delay"
                    % (n_total, args.cores_per_gpu, args.neurons_per_core, args.out_degree, 100 * args.p_fire))
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
    multi = world > 1 or args.force_dist
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

    # Device warm-up, NOT part of the workload: the first ~30 ms of load after an idle period run at unsettled clocks
    # (the same steps repeated after a reset are 10-25 % faster, profiles/r02_step_profile.txt).  Simulate, then put the
    # chip back into its initial state, so the W warm-up steps and the K timed steps are steps 1..W+K of the simulation.
    device_warmup = args.device_warmup if args.workload == "c3" and args.timing == "simple" else 0
    if device_warmup > 0:
        run_steps(device_warmup)
        sync()
        chip.reset()
    run_steps(args.warmup)
    sync()
    if dist:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    run_data = run_steps(args.steps)
    sync()
    if dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    agg = {k: float(run_data[k]) for k in ("spikes", "packets_sent", "neurons_updated", "neurons_fired")}
    if dist:
        import torch
        tmax = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax[0])

    # ---- roofline pass: HIP events around every kernel on its own stream (one rank, the chip's own stream) ----
    roof = None
    if args.timed_steps > 0 and args.timing == "simple" and not multi:
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
        # (1) the contract's figure -- SURVEY 8(d): 48 B per neuron update, 28 B per synaptic event, 96 B per message.
        #     The delivery kernel does the per-event and per-message work, the neuron kernel the per-neuron work.
        deliver_bytes = 28.0 * ev + 96.0 * msgs
        achieved = deliver_bytes / (dm.value * 1e-3) / 1e9 if dm.value > 0 else 0.0
        # (2) what THIS design has to move per launch, from the chip's own layout (sanafe_hip_layout_bytes):
        #     delivery = synapse words + axon records + chunk tables + slice descriptors + spike bitmap, each read once
        #     when every chunk is streamed (an upper bound when few axons spike), + one 17-byte write-back per neuron;
        #     neuron launch = per-slot state read + written, + 40 B per fired neuron.
        lay = (C.c_uint64 * 8)()
        H.sanafe_hip_layout_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        if H.sanafe_hip_layout_bytes(dev, lay, 8) != 0:
            raise RuntimeError(H.sanafe_hip_last_error().decode())
        lay = [int(x) for x in lay]
        design_deliver = float(sum(lay[0:5])) + 17.0 * upd
        design_neuron = float(lay[5] + lay[6]) + lay[7] * fired
        roof = {"bound": "hbm", "kernel": "deliver_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_launch": deliver_bytes, "avg_launch_ms": dm.value,
                "design_bytes_per_launch": design_deliver,
                "design_bytes_parts": {"synapse_words": lay[0], "axon_records": lay[1], "chunk_tables": lay[2],
                                       "slice_descriptors": lay[3], "spike_bitmap": lay[4], "write_back": 17.0 * upd},
                "frac_design": design_deliver / (dm.value * 1e-3) / 1e9 / HBM_PEAK_GBS if dm.value > 0 else 0.0,
                "neuron_kernel": {"avg_launch_ms": nm.value, "algorithmic_bytes_per_launch": 48.0 * upd,
                                  "achieved_GBps": (48.0 * upd) / (nm.value * 1e-3) / 1e9 if nm.value > 0 else 0.0,
                                  "frac": (48.0 * upd) / (nm.value * 1e-3) / 1e9 / HBM_PEAK_GBS if nm.value > 0 else 0.0,
                                  "design_bytes_per_launch": design_neuron,
                                  "frac_design": design_neuron / (nm.value * 1e-3) / 1e9 / HBM_PEAK_GBS if nm.value > 0 else 0.0},
                "reduce_kernel_avg_ms": rm.value, "launches": ln.value,
                "whole_step": {"algorithmic_bytes": 48.0 * upd + deliver_bytes,
                               "achieved_GBps": (48.0 * upd + deliver_bytes) / ((nm.value + dm.value + rm.value) * 1e-3) / 1e9}}

    if roof is not None:
        # (3) HBM bytes per launch as the PMC counters saw them on this very workload (profiles/collect.sh,
        #     summarize.py: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, plus WRITE_SIZE; separate
        #     --pmc passes, so they come from an earlier run of the same command -- source and date are given).
        import glob
        newest = -1
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "summary.json"))):
            try:
                with open(path) as f:
                    summ = json.load(f)
                if summ["bench_line"]["config"]["workload"] == workload and "hbm_bytes_per_launch" in summ["kernels"]["deliver_kernel"] \
                        and summ.get("collected_unix", 0) > newest:
                    newest = summ.get("collected_unix", 0)
                    roof["traffic"] = summ["kernels"]["deliver_kernel"]["hbm_bytes_per_launch"]
                    roof["traffic_source"] = os.path.relpath(path, ROOT)
                    roof["traffic_collected_utc"] = time.strftime("%Y-%m-%d %H:%M", time.gmtime(newest)) if newest > 0 else None
                    nk = summ["kernels"].get("neuron_kernel", {})
                    if "hbm_bytes_per_launch" in nk:
                        roof["neuron_kernel"]["traffic"] = nk["hbm_bytes_per_launch"]
            except (KeyError, TypeError, ValueError, OSError):
                continue
        if roof["traffic"]:
            roof["traffic_GBps"] = roof["traffic"] / (roof["avg_launch_ms"] * 1e-3) / 1e9
            roof["traffic_frac_of_peak"] = roof["traffic_GBps"] / HBM_PEAK_GBS
        roof["note"] = ("achieved/frac price the launch with SURVEY 8(d)'s byte model (28 B per synaptic event + 96 B per message), "
                        "which assumes materialised 40-byte messages and 12-byte synapses"
                        + ("; frac > 1 because this design never materialises messages and packs a synapse in %.1f bytes, so it moves "
                           "%.1fx fewer bytes than the model" % (lay[0] / max(1.0, float(info["n_synapses"])), deliver_bytes / max(1.0, design_deliver))
                           if roof["frac"] > 1.0 else "")
                        + "; frac_design prices the same event-timed launch with the bytes of the chip's own layout "
                          "(design_bytes_parts); traffic* are the HBM bytes the PMC counters measured for this workload "
                          "(traffic_source, traffic_collected_utc) over the event-timed duration of this run")

    cpu = None
    if rank == 0 and not multi and not args.no_cpu_baseline and args.workload == "c3" and args.timing == "simple":
        c = cpu_baseline(S, args)
        events_per_step = agg["spikes"] / args.steps
        est = c["events_per_s"] / events_per_step if events_per_step > 0 else c["steps_per_s"]
        cpu = {"value": est, "unit": "timesteps/s", "cores": 1, "kind": "port", "host": host_info(),
               "sample": c["sample"] + "; value = measured synaptic-events/s (%.3g) / events per step of the GPU workload (%.3g)"
                         % (c["events_per_s"], events_per_step),
               "measured": {"timesteps_per_s_on_sample": c["steps_per_s"], "synaptic_events_per_s": c["events_per_s"],
                            "neuron_updates_per_s": c["updates_per_s"], "timesteps_per_s_windows": c["all_steps_per_s"]}}

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
                              "note": "clock warm-up before the W warm-up steps; the timed steps are steps W+1..W+K of a fresh simulation"},
            "roofline": roof, "cpu_baseline": cpu,
        }
        real_stdout.write(json.dumps(out) + "\n")
        real_stdout.flush()
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
