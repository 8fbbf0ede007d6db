"""Network builders shared by the tests, bench.py and __graft_entry__.smoke().

All inputs come from committed fixtures (tests/golden/) or seeded generators; nothing here reads
/root/reference, which does not exist on the GPU box.
"""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def example(S):
    """Config C1: arch/example_chip.yaml + snn/example_snn.yaml."""
    arch = S.presets.example_chip(api=S.description)
    return arch, example_snn(S, arch)


def tutorial5_dvs(S, core_counts=(1, 4, 16, 16, 4, 1)):
    """The network tutorial/tutorial_5_dvs.ipynb builds with sanafe.layers (sanafe/layers.py:85-330)
    from sanafe/examples/dvs_challenge.npz on the Loihi architecture; the notebook asserts
    results["neurons_fired"] == 365277 after chip.sim(1000)."""
    arch = S.presets.loihi(api=S.description)
    d = np.load(os.path.join(GOLDEN, "dvs_challenge.npz"))
    th = d["thresholds"]
    net = S.description.Network()
    g0 = net.create_neuron_group("input_0", 32 * 32, {"threshold": th[0]})
    layers = [(g0, 32, 32, 1)]
    for i, (name, stride) in enumerate((("conv1", 2), ("conv2", 1), ("conv3", 1), ("conv4", 1))):
        w = d[name]
        kw, kh, cin, cout = w.shape
        pg, pw, ph, pc = layers[-1]
        ow, oh = 1 + (pw - kw) // stride, 1 + (ph - kh) // stride
        g = net.create_neuron_group("conv2d_%d" % i, ow * oh * cout, {"threshold": th[i + 1]})
        pg.connect_neurons_conv2d(g, {"w": w.flatten()}, pw, ph, pc, kw, kh, cout, stride, stride)
        layers.append((g, ow, oh, cout))
    g = net.create_neuron_group("dense_0", 11, {"threshold": th[5]})
    layers[-1][0].connect_neurons_dense(g, {"w": d["dense1"].flatten()})
    layers.append((g, 11, 1, 1))
    # `n.set_attributes(model_attributes={"bias": b})`: numpy float64 -> float32-narrowed double
    g0.set_attribute_column("bias", d["inputs"].astype(np.float32).astype(np.float64), S.description.ATTR_DOUBLE)
    cores = arch.cores()
    g0.map_to_core(arch.tiles[0].cores[0])  # the notebook maps layer0 once before map_layer_to_cores
    k = 0
    for (grp, _, _, _), cc in zip(layers, core_counts):
        n, per = len(grp), len(grp) // cc
        for idx in range(cc):
            lo, hi = idx * per, (len(grp) if idx == cc - 1 else (idx + 1) * per)
            grp.map_to_core(cores[k], lo, hi)
            k += 1
    return arch, net


def dvs_yaml(S):
    """Config C2: arch/loihi.yaml + snn/dvs.yaml, rebuilt from tests/golden/dvs_yaml.npz."""
    D, Y = S.description, S.yaml_io
    arch = S.presets.loihi(api=S.description)
    f = np.load(os.path.join(GOLDEN, "dvs_yaml.npz"))
    meta = json.loads(bytes(f["meta"]).decode())
    net = S.description.Network(meta["name"])
    for g in meta["groups"]:
        a = dict(g["attributes"])
        typed = {}
        for k, v in a.items():
            if k in Y.SKIP_KEYS:
                continue
            s = str(v).lower() if isinstance(v, bool) else str(v)
            typed[k] = (Y._scalar(s), D.FWD_ALL)
        grp = net.create_neuron_group(g["name"], g["count"], None, a.get("synapse_hw_name", ""),
                                      a.get("dendrite_hw_name", ""), bool(a.get("log_potential", False)),
                                      bool(a.get("log_spikes", False)), a.get("soma_hw_name", ""), _typed_attrs=typed)
        key = "bias_" + g["name"]
        if key in f:
            grp.set_attribute_column("bias", f[key], D.ATTR_INT)
    for i, e in enumerate(meta["edges"]):
        sg, tg = (x.strip() for x in e["desc"].split("->"))
        p = dict(e["params"])
        t = p.pop("type")
        w = f["w_%d" % i].astype(np.float64)
        if t == "conv2d":
            net[sg].connect_neurons_conv2d(net[tg], {"weight": w}, narrow_float=False, **p)
        else:
            net[sg].connect_neurons_dense(net[tg], {"weight": w}, narrow_float=False)
    for addr, core in meta["mappings"]:
        gname, off = addr.split(".")
        t, c = (int(x) for x in core.split("."))
        net[gname].map_to_core(arch.tiles[t].cores[c], int(off), int(off) + 1)
    return arch, net


def random_loihi(S, n_tiles=4, neurons_per_core=64, cores_used=None, out_degree=32, p_fire=0.1, seed=1,
                 arch_kind="large", delays=False, weights="int", refractory=False, dendrite=None):
    """Config C3-style synthetic SNN (recipe after scripts/tcad2025/random_network.py:63-105):
    LIF neurons, threshold 64, reset 0, force_update, a fraction p_fire with bias 128 (fire every step),
    fixed out-degree with targets drawn without replacement, integer weights in {-8..8}\\{0}."""
    D = S.description
    rng = np.random.default_rng(seed)
    if arch_kind == "before_dendrite":
        # buffer_position: dendrite, buffer_inside_unit: false -- only the last event of a step reaches the accumulator
        w = max(1, int(np.ceil(np.sqrt(n_tiles))))
        arch = S.presets.loihi_large(n_tiles=n_tiles, width=w, height=int(np.ceil(n_tiles / w)), n_inputs=4, api=S.description,
                                     buffer_inside_unit=False)
        dend = "loihi_dendrites"
    elif arch_kind == "large":
        w = max(1, int(np.ceil(np.sqrt(n_tiles))))
        arch = S.presets.loihi_large(n_tiles=n_tiles, width=w, height=int(np.ceil(n_tiles / w)), n_inputs=4, api=S.description)
        dend = "loihi_dendrites_delay"  # quirk 1: plain accumulator loses all input inside the dendrite
    else:
        arch = S.presets.loihi(n_inputs=4, api=S.description)
        arch.tiles = arch.tiles[:n_tiles]
        arch._cores = arch._cores[:4 * n_tiles]
        dend = "loihi_dendrites"
    dend = dendrite or dend
    cores = arch.cores()
    cores_used = cores_used or len(cores)
    n = cores_used * neurons_per_core
    net = S.description.Network("random")
    attrs = {"threshold": 64, "reset": 0, "force_update": True}
    if refractory:
        attrs["refractory_delay"] = 2
    g = net.create_neuron_group("n", n, attrs, "loihi_sparse_synapse", dend, False, True, "loihi_lif")
    bias = np.where(rng.random(n) < p_fire, 128, 0).astype(np.int64)
    g.set_attribute_column("bias", bias, D.ATTR_INT)
    out_degree = min(out_degree, n)
    src = np.repeat(np.arange(n, dtype=np.int64), out_degree)
    dst = np.empty(n * out_degree, dtype=np.int64)
    for i in range(n):
        dst[i * out_degree:(i + 1) * out_degree] = rng.choice(n, size=out_degree, replace=False)
    if weights == "int":
        w = rng.integers(1, 9, size=len(src)) * rng.choice([-1, 1], size=len(src))
    elif weights == "int12":
        w = rng.integers(-2000, 2001, size=len(src))
    else:
        w = rng.normal(size=len(src)) * 4.0
    at = {"weight": w.astype(np.float64)}
    if delays:
        at["delay"] = rng.integers(0, 6, size=len(src))
    g.connect_neurons_sparse(g, at, np.stack([src, dst], axis=1), narrow_float=False)
    for c in range(cores_used):
        g.map_to_core(cores[c], c * neurons_per_core, (c + 1) * neurons_per_core)
    return arch, net


def c3_delivery_shape(S, cores=512, neurons_per_core=512, dest_cores=8, out_degree=41, p_fire=0.1, seed=1, weights="int",
                      delays=True):
    """The per-core DELIVERY shape of config C3 at a size the oracle can follow: all `cores` x `neurons_per_core` source
    neurons of the bench recipe (512 x 512 = 262,144), but every neuron draws its `out_degree` targets among the neurons
    of `dest_cores` cores only.  Each of those cores then sees what a core of the full network sees: ~260 k inbound axons
    with ~5 synapses each from every source core of the chip (16+ delivery slices of >= 8,192 axons with
    SANAFE_MIN_SLICE_AXONS=16384, runs of 8 chunks, write-back shared between the slices) -- with 10.7 M synapses
    instead of 687 M."""
    D = S.description
    rng = np.random.default_rng(seed)
    n_tiles = (cores + 3) // 4
    w = max(1, int(np.ceil(np.sqrt(n_tiles))))
    arch = S.presets.loihi_large(n_tiles=n_tiles, width=w, height=int(np.ceil(n_tiles / w)), n_inputs=4, api=S.description)
    n = cores * neurons_per_core
    net = D.Network("c3_shape")
    g = net.create_neuron_group("n", n, {"threshold": 64, "reset": 0, "force_update": True}, "loihi_sparse_synapse",
                                "loihi_dendrites_delay", False, True, "loihi_lif")
    g.set_attribute_column("bias", np.where(rng.random(n) < p_fire, 128, 0).astype(np.int64), D.ATTR_INT)
    # destination cores spread over the chip (first, last and in between), so sources come from lower AND higher core ids
    dcores = np.unique(np.linspace(0, cores - 1, dest_cores).astype(np.int64))
    pool = (dcores[:, None] * neurons_per_core + np.arange(neurons_per_core)[None, :]).ravel()
    src = np.repeat(np.arange(n, dtype=np.int64), out_degree)
    dst = pool[rng.integers(0, len(pool), size=len(src))]  # (repeated targets are separate synapses of the same axon)
    if weights == "int":
        wv = (rng.integers(1, 9, size=len(src)) * rng.choice([-1, 1], size=len(src))).astype(np.float64)
        # keep the targets near threshold instead of saturated: ~27 k spiking sources x 41 / 4096 targets = 270 events each
        wv = np.where(rng.random(len(src)) < 0.5, wv, -wv)
    else:
        wv = rng.normal(size=len(src)) * 4.0
    at = {"weight": wv}
    if delays:
        at["delay"] = rng.integers(0, 6, size=len(src))
    g.connect_neurons_sparse(g, at, np.stack([src, dst], axis=1), narrow_float=False)
    ac = arch.cores()
    for c in range(cores):
        g.map_to_core(ac[c], c * neurons_per_core, (c + 1) * neurons_per_core)
    return arch, net


def truenorth_net(S, n_tiles=16, neurons_per_core=256, remote_fraction=0.8, seed=1):
    """Config C4-style synthetic SNN (after scripts/tcad2025/compare_nemo_perf.py:52-101):
    `truenorth` neurons, threshold 0, reset -1, leak 0, force_update, weight 1, one out-edge per neuron,
    80 % of them to another core."""
    rng = np.random.default_rng(seed)
    w = max(1, int(np.ceil(np.sqrt(n_tiles))))
    arch = S.presets.truenorth(n_tiles=n_tiles, width=w, height=int(np.ceil(n_tiles / w)), api=S.description)
    cores = arch.cores()
    n = n_tiles * neurons_per_core
    net = S.description.Network("tn")
    g = net.create_neuron_group("tn", n, {"threshold": 0, "reset": -1, "leak": 0, "force_update": True},
                                "core_synapses", "core_dendrites", False, True, "core_soma")
    src = np.arange(n, dtype=np.int64)
    core_of = src // neurons_per_core
    remote = rng.random(n) < remote_fraction
    dst_core = np.where(remote, (core_of + 1 + rng.integers(0, max(1, n_tiles - 1), size=n)) % n_tiles, core_of)
    dst = dst_core * neurons_per_core + rng.integers(0, neurons_per_core, size=n)
    g.connect_neurons_sparse(g, {"weight": np.ones(n)}, np.stack([src, dst], axis=1), narrow_float=False)
    for c in range(n_tiles):
        g.map_to_core(cores[c], c * neurons_per_core, (c + 1) * neurons_per_core)
    return arch, net


def stochastic(S, tmp_path, n_lif=96, n_in=6, n_tn=0, seed=9, noise_bits=None, silent_inputs=()):
    """Every sequential host-side source the models consume: Poisson inputs
    (std::mt19937 per input unit), LIF neurons on a soma unit that reads a noise file, spread over several cores."""
    D = S.description
    rng = np.random.default_rng(seed)
    noise = os.path.join(str(tmp_path), "noise.csv")
    vals = rng.integers(0, 512, size=157)
    with open(noise, "w") as f:
        for i, v in enumerate(vals):
            f.write("%d\n" % v if i != 40 else "oops\n")  # an entry that does not parse reads as 0
    arch = S.presets.loihi(n_inputs=8, api=S.description, noise_file=noise, noise_bits=noise_bits)
    cores = arch.cores()
    net = D.Network("stoch")
    gin = net.create_neuron_group("in", n_in, {}, "loihi_sparse_synapse", "loihi_dendrites", False, True)
    for i in range(n_in):
        attrs = {"poisson": ((D.ATTR_DOUBLE, 0.15 + 0.1 * i, None, None), D.FWD_ALL)}
        if i in silent_inputs:
            attrs = {}  # no Poisson rate at load(): the unit's generator draws at every update all the same
        if i == 1:
            attrs["spikes"] = ((D.ATTR_LIST, 0.0, None, [0.0, 1.0, 1.0, 0.0, 1.0]), D.FWD_ALL)
        gin.apply_config(i, i + 1, soma_hw_name="loihi_inputs[%d]" % (i % 3), attrs=attrs)
    g = net.create_neuron_group("lif", n_lif, {"threshold": 90, "reset": 0, "leak_decay": 0.9, "refractory_delay": 1},
                                "loihi_sparse_synapse", "loihi_dendrites", True, True, "loihi_stochastic_lif")
    plain = net.create_neuron_group("plain", 16, {"threshold": 20, "reset": 0}, "loihi_sparse_synapse", "loihi_dendrites",
                                    False, True, "loihi_lif")
    pairs = np.stack([rng.integers(0, n_in, size=400), rng.integers(0, n_lif, size=400)], axis=1)
    gin.connect_neurons_sparse(g, {"weight": rng.integers(1, 30, size=400).astype(np.float64)}, pairs, narrow_float=False)
    pairs = np.stack([rng.integers(0, n_lif, size=300), rng.integers(0, 16, size=300)], axis=1)
    g.connect_neurons_sparse(plain, {"weight": rng.integers(1, 9, size=300).astype(np.float64)}, pairs, narrow_float=False)
    # inputs: two per core on cores 0..2 (each input unit holds one neuron); stochastic LIF over 3 cores
    for i in range(n_in):
        gin.map_to_core(cores[i // 3 + 5 * (i % 3 == 2)], i, i + 1)
    third = n_lif // 3
    g.map_to_core(cores[1], 0, third)
    g.map_to_core(cores[0], third, 2 * third)
    g.map_to_core(cores[7], 2 * third, n_lif)
    plain.map_to_core(cores[1], 0, 16)
    return arch, net


def stochastic_truenorth(S, n_tiles=6, neurons_per_core=64, seed=2):
    """TrueNorth neurons whose threshold test adds `std::rand() & random_mask` (src/models.cpp:745-759)."""
    D = S.description
    arch, net = truenorth_net(S, n_tiles=n_tiles, neurons_per_core=neurons_per_core, seed=seed)
    g = net._order[0]
    rng = np.random.default_rng(seed)
    n = g.count
    mask = np.where(rng.random(n) < 0.6, rng.choice([1, 3, 7, 15, 255], size=n), 0).astype(np.int64)
    g.set_attribute_column("random_mask", mask, D.ATTR_INT)
    g.set_attribute_column("threshold", rng.integers(2, 12, size=n).astype(np.int64), D.ATTR_INT)
    g.set_attribute_column("leak", np.ones(n, dtype=np.int64), D.ATTR_INT)
    g.set_attribute_column("leak_towards_zero", np.zeros(n, dtype=np.int64), D.ATTR_INT)
    return arch, net


def taps_dendrites(S, n=24, n_in=10, seed=3, max_taps=5, buffer=("soma", False), neurons_per_unit=1):
    """`taps` dendrites (MultiTapModel1D) in the style of arch/demo_with_dendrites.yaml + snn/dendrite.yaml: one
    dendrite unit per neuron, 1..5 taps with their own time/space constants, synapses that name their tap, inputs
    that spike from trains, recurrent edges between the dendritic neurons."""
    D = S.description
    rng = np.random.default_rng(seed)
    arch = D.Architecture("dendrite", 2, 1, 4)
    for t in range(2):
        tile = arch.create_tile("tile[%d]" % t, energy_north_hop=2.0e-12, latency_north_hop=1.4e-9, energy_east_hop=2.5e-12,
                                latency_east_hop=1.2e-9, energy_south_hop=2.0e-12, latency_south_hop=1.5e-9,
                                energy_west_hop=1.8e-12, latency_west_hop=2.0e-9)
        core = arch.create_core("core[0]", tile.id, buffer[0], buffer[1], 100)
        core.create_axon_in("axon_in", 0.0, 0.0)
        core.create_synapse("synapse", "current_based", {"energy_process_spike": 20.0e-12, "latency_process_spike": 3.0e-9})
        for i in range(n):
            core.create_dendrite("dendrite[%d]" % i, "taps", {"energy_update": 1.0e-12, "latency_update": 0.5e-9})
        core.create_dendrite("plain", "accumulator", {"energy_update": 1.0e-12, "latency_update": 0.5e-9})
        core.create_soma("soma", "leaky_integrate_fire",
                         {"energy_access_neuron": 20.0e-12, "latency_access_neuron": 3.0e-9, "energy_update_neuron": 10.0e-12,
                          "latency_update_neuron": 1.0e-9, "energy_spike_out": 60.0e-12, "latency_spike_out": 30.0e-9})
        for i in range(n_in):
            core.create_soma("dummy_input[%d]" % i, "input",
                             {"energy_access_neuron": 0.0, "latency_access_neuron": 0.0, "energy_update_neuron": 0.0,
                              "latency_update_neuron": 0.0, "energy_spike_out": 0.0, "latency_spike_out": 0.0})
        core.create_axon_out("axon_out", 100.0e-12, 5.0e-9)
    cores = arch.cores()
    net = D.Network("dendrites")
    gin = net.create_neuron_group("inputs", n_in, {}, "synapse", "plain", False, True)
    for i in range(n_in):
        train = [float(x) for x in (rng.random(40) < 0.35)]
        gin.apply_config(i, i + 1, soma_hw_name="dummy_input[%d]" % i,
                         attrs={"spikes": ((D.ATTR_LIST, 0.0, None, train), D.FWD_ALL)})
    g = net.create_neuron_group("dendrite", n, {"threshold": 12, "reset": 0, "leak_decay": 0.9}, "synapse", "", True, True,
                                "soma")
    taps = rng.integers(1, max_taps + 1, size=n)
    units = max(1, (n // 2) // neurons_per_unit)  # unit instances per core: neurons i and i + units of a core share one
    if neurons_per_unit > 1:
        taps = taps[np.arange(n) % units]          # one RC line: the unit keeps the configuration applied last anyway
    for i in range(n):
        k = int(taps[i])
        attrs = {"taps": ((D.ATTR_INT, float(k), None, None), D.FWD_ALL),
                 "time_constants": ((D.ATTR_LIST, 0.0, None, [float(x) for x in rng.choice([0.5, 0.75, 0.875], size=k)]), D.FWD_ALL),
                 "space_constants": ((D.ATTR_LIST, 0.0, None, [float(x) for x in rng.choice([0.125, 0.25], size=max(k - 1, 0))]),
                                     D.FWD_ALL)}
        g.apply_config(i, i + 1, dendrite_hw_name="dendrite[%d]" % (i % units), attrs=attrs)
    # inputs -> dendritic neurons, each synapse naming a tap of its target
    pairs = np.stack([rng.integers(0, n_in, size=90), rng.integers(0, n, size=90)], axis=1)
    gin.connect_neurons_sparse(g, {"weight": rng.integers(2, 9, size=90).astype(np.float64),
                                   "tap": [int(rng.integers(0, taps[d])) for d in pairs[:, 1]]}, pairs, narrow_float=False)
    pairs = np.stack([rng.integers(0, n, size=60), rng.integers(0, n, size=60)], axis=1)
    g.connect_neurons_sparse(g, {"weight": rng.integers(1, 5, size=60).astype(np.float64),
                                 "tap": [int(rng.integers(0, taps[d])) for d in pairs[:, 1]]}, pairs, narrow_float=False)
    half = n // 2
    g.map_to_core(cores[0], 0, half)
    g.map_to_core(cores[1], half, n)
    gin.map_to_core(cores[0], 0, n_in // 2)
    gin.map_to_core(cores[1], n_in // 2, n_in)
    return arch, net


def shared_input_units(S, seed=8):
    """Several neurons on ONE `input` unit share its spike train, cursor and Poisson generator (as the three inputs of
    snn/dendrite.yaml do on `dummy_input`): every update consumes the next element / draw."""
    D = S.description
    rng = np.random.default_rng(seed)
    arch = S.presets.loihi(n_inputs=4, api=S.description)
    cores = arch.cores()
    net = D.Network("shared_inputs")
    gin = net.create_neuron_group("in", 9, {}, "loihi_sparse_synapse", "loihi_dendrites", False, True)
    unit_of = [0, 0, 0, 1, 1, 2, 0, 0, 3]      # per neuron: loihi_inputs[k] of its core
    core_of = [0, 0, 0, 0, 0, 0, 1, 1, 1]
    for i in range(9):
        attrs = {}
        if i in (0, 2, 4, 6):
            attrs["spikes"] = ((D.ATTR_LIST, 0.0, None, [float(x) for x in (rng.random(40) < 0.5)]), D.FWD_ALL)
        if i in (1, 7):
            attrs["poisson"] = ((D.ATTR_DOUBLE, 0.25 + 0.25 * (i == 7), None, None), D.FWD_ALL)
        if i == 3:
            attrs["rate"] = ((D.ATTR_DOUBLE, 0.25, None, None), D.FWD_ALL)
        gin.apply_config(i, i + 1, soma_hw_name="loihi_inputs[%d]" % unit_of[i], attrs=attrs)
    g = net.create_neuron_group("lif", 40, {"threshold": 30, "reset": 0, "leak_decay": 0.875}, "loihi_sparse_synapse",
                                "loihi_dendrites", False, True, "loihi_lif")
    pairs = np.stack([rng.integers(0, 9, size=200), rng.integers(0, 40, size=200)], axis=1)
    gin.connect_neurons_sparse(g, {"weight": rng.integers(1, 12, size=200).astype(np.float64)}, pairs, narrow_float=False)
    # mapping order decides both whose attributes the unit keeps and the update order
    for i in (2, 0, 1, 4, 3, 5, 8, 6, 7):
        gin.map_to_core(cores[core_of[i]], i, i + 1)
    g.map_to_core(cores[2], 0, 40)
    return arch, net


def relay_plugin_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "plugins", "librelay_units.so")


def host_cores(S, position="soma_inside", plugin_units=False, builtin_twin=False, n_per_core=48, out_degree=14, seed=5,
               soma="truenorth"):
    """A 6-core chip whose cores 2 and 4 cannot run on the device (mapper.hpp: MappedChip::HostCore) next to cores that can:
      position "soma_inside" / "axon_out": `buffer_position: soma` with the buffer inside the unit / `axon_out` on those
               cores -- the soma is called once per synaptic event (src/mapped.cpp:27-58);
      plugin_units: those cores keep `buffer_position: soma` (outside) but their synapse and dendrite units are plugins
               (tests/plugins/relay_units.cpp: current_based / accumulator arithmetic behind the plugin interface);
               builtin_twin=True builds the SAME chip on the built-in units instead (what the oracle and the device run).
    TrueNorth somas everywhere (the built-in soma that tolerates several updates per step), non-zero costs on every
    unit, integer weights, traffic in every direction between host and device cores."""
    D = S.description
    rng = np.random.default_rng(seed)
    arch = D.Architecture("mixed", 3, 2, 4, {0: 0.0, 2: 0.9e-6})
    syn_costs = {"energy_process_spike": 20.0e-12, "latency_process_spike": 3.0e-9}
    dend_costs = {"energy_update": 7.0e-12, "latency_update": 1.5e-9}
    soma_costs = {"energy_access_neuron": 20.0e-12, "latency_access_neuron": 3.0e-9, "energy_update_neuron": 10.0e-12,
                  "latency_update_neuron": 1.0e-9, "energy_spike_out": 60.0e-12, "latency_spike_out": 30.0e-9}
    for t in range(6):
        tile = arch.create_tile("tile[%d]" % t, energy_north_hop=2.0e-12, latency_north_hop=1.4e-9, energy_east_hop=2.5e-12,
                                latency_east_hop=1.2e-9, energy_south_hop=2.0e-12, latency_south_hop=1.5e-9,
                                energy_west_hop=1.8e-12, latency_west_hop=2.0e-9)
        special = t in (2, 4)
        if special and not plugin_units and position == "soma_inside":
            core = arch.create_core("core[0]", tile.id, "soma", True, 256)
        elif special and not plugin_units and position == "axon_out":
            core = arch.create_core("core[0]", tile.id, "axon_out", False, 256)
        else:
            core = arch.create_core("core[0]", tile.id, "soma", False, 256)
        core.create_axon_in("axon_in", 1.0e-12, 2.0e-9)
        if special and plugin_units and not builtin_twin:
            core.create_synapse("syn", "test_synapse", dict(syn_costs), plugin=relay_plugin_path())
            core.create_dendrite("dend", "test_dendrite", dict(dend_costs), plugin=relay_plugin_path())
        else:
            core.create_synapse("syn", "current_based", dict(syn_costs))
            core.create_dendrite("dend", "accumulator", dict(dend_costs))
        core.create_soma("soma", "truenorth" if soma == "truenorth" else "leaky_integrate_fire", dict(soma_costs))
        core.create_axon_out("axon_out", 100.0e-12, 5.0e-9)
    cores = arch.cores()
    n = 6 * n_per_core
    net = D.Network("mixed")
    g = net.create_neuron_group("n", n, {"reset": 0, "leak": 1}, "syn", "dend", False, True, "soma")
    g.set_attribute_column("threshold", rng.integers(6, 40, size=n).astype(np.int64), D.ATTR_INT)
    g.set_attribute_column("bias", np.where(rng.random(n) < 0.35, rng.integers(2, 9, size=n), 0).astype(np.int64), D.ATTR_INT)
    src = np.repeat(np.arange(n, dtype=np.int64), out_degree)
    dst = rng.integers(0, n, size=len(src))
    w = (rng.integers(1, 6, size=len(src)) * rng.choice([-1, 1, 1], size=len(src))).astype(np.float64)
    g.connect_neurons_sparse(g, {"weight": w}, np.stack([src, dst], axis=1), narrow_float=False)
    for c in range(6):
        g.map_to_core(cores[c], c * n_per_core, (c + 1) * n_per_core)
    return arch, net


def hh_plugin_path():
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "plugins", "libhodgkin_huxley.so")


def hodgkin_huxley(S, k=12, lif=24, seed=4, spread=False):
    """Config C5: `k` Hodgkin-Huxley plugin somas (snn/hh_example.net scaled up: m=0.0529 n=0.3177
    h=0.5961, a distinct `current` each) driving a small LIF population on the Loihi architecture."""
    D = S.description
    rng = np.random.default_rng(seed)
    arch = S.presets.loihi_with_plugin_somas(k, hh_plugin_path(), api=S.description)
    cores = arch.cores()
    net = S.description.Network("hh")
    g = net.create_neuron_group("hh", k, {"m": 0.0529, "n": 0.3177, "h": 0.5961}, "loihi_sparse_synapse", "", True, True)
    for i in range(k):
        g.apply_config(i, i + 1, soma_hw_name="hh[%d]" % i,
                       attrs={"current": ((D.ATTR_DOUBLE, float(25 * i), None, None), D.FWD_ALL),
                              "m": ((D.ATTR_DOUBLE, 0.0529, None, None), D.FWD_ALL),
                              "n": ((D.ATTR_DOUBLE, 0.3177, None, None), D.FWD_ALL),
                              "h": ((D.ATTR_DOUBLE, 0.5961, None, None), D.FWD_ALL)})
    out = net.create_neuron_group("lif", lif, {"threshold": 3, "reset": 0, "leak_decay": 0.9}, "loihi_sparse_synapse", "",
                                  True, True, "loihi_lif")
    pairs = np.array([(i, j) for i in range(k) for j in rng.choice(lif, size=6, replace=False)])
    g.connect_neurons_sparse(out, {"weight": rng.integers(1, 4, size=len(pairs)).astype(np.float64)}, pairs, narrow_float=False)
    ring = np.array([(i, (i + 1) % k) for i in range(k)])
    g.connect_neurons_sparse(g, {"weight": np.ones(k)}, ring, narrow_float=False)
    if spread:  # plugin somas on both halves of the chip's tiles (tile-sharded runs: both ranks hold some)
        g.map_to_core(cores[0], 0, k // 2)
        g.map_to_core(cores[70], k // 2, k)
        out.map_to_core(cores[5], 0, lif // 2)
        out.map_to_core(cores[66], lif // 2, lif)
        return arch, net
    g.map_to_core(cores[0])
    out.map_to_core(cores[5], 0, lif // 2)
    out.map_to_core(cores[0], lif // 2, lif)
    return arch, net


def example_snn(S, arch):
    """snn/example_snn.yaml restated with the Python twin's typed setters, so that the description is
    identical to what the YAML front-end reads (4 neurons, 5 synapses)."""
    D = S.description
    net = D.Network("example_snn")
    gin = net.create_neuron_group("in", 2, log_spikes=True)
    gin.apply_config(0, 1, log_spikes=False)
    gin.set_attribute("spikes", (D.ATTR_LIST, 0.0, None, [1, 0, 1]), D.FWD_ALL, 1, 2)
    gout = net.create_neuron_group("out", 2)
    gout.apply_config(0, 2, log_potential=True,
                      attrs={"threshold": ((D.ATTR_INT, 2.0, None, None), D.FWD_SOMA),
                             "log_u": ((D.ATTR_BOOL, 1.0, None, None), D.FWD_ALL)})
    import numpy as np
    net._add_edges(np.array([gout.base + 1]), np.array([gout.base + 1]), np.array([-4.0]))
    gin.connect_neurons_dense(gout, {"weight": np.array([-1.0, 2.0, 1.0, 3.0])}, narrow_float=False)
    cores = arch.tile_cores(0)
    gin.apply_config(0, 1, soma_hw_name="demo_input")
    gin.map_to_core(cores[0], 0, 1)
    gin.apply_config(1, 2, soma_hw_name="demo_input")
    gin.map_to_core(cores[1], 1, 2)
    gout.map_to_core(cores[0])
    return net
