"""Parity of the MI355X path (through the C ABI) against the CPU oracle on the same inputs.

Integer outputs (statuses, spike lists, counters, message ids/hops) are compared exactly;
potentials bit-for-bit where weights are integers (all reference configs) and within 1e-9
relative otherwise; energies and simulated time within 1e-9 relative (north_star asks 1e-6)."""
import os

import numpy as np
import pytest

import nets
from oracle.oracle import OracleChip

pytestmark = pytest.mark.gpu

INT_KEYS = (("spikes", "spike_count"), ("packets_sent", "packets_sent"), ("neurons_updated", "neurons_updated"),
            ("neurons_fired", "neurons_fired"), ("total_hops", "total_hops"))
DBL_KEYS = ("total_energy", "synapse_energy", "dendrite_energy", "soma_energy", "network_energy", "sim_time")
REL = 1e-9


def make(S, arch, net):
    chip = S.SpikingChip(arch)
    chip.load(net)
    return chip, OracleChip(S.to_desc(arch, net))


def check_stepwise(S, arch, net, steps, timing="simple", exact_v=True):
    chip, orc = make(S, arch, net)
    for t in range(steps):
        a = chip.run(1, timing, record=True)
        b = orc.step(timing)
        for ka, kb in INT_KEYS:
            assert a[ka] == b[kb], (t, ka, a[ka], b[kb])
        for k in DBL_KEYS:
            assert a[k] == pytest.approx(b[k], rel=REL, abs=1e-30), (t, k)
        st = orc.status()
        assert np.array_equal(chip.status(), st), t
        assert np.array_equal(chip.step_fired(0), (st == 3).astype(np.uint8)), t
        if exact_v:
            assert np.array_equal(chip.potentials(), orc.potentials()), t
        else:
            assert np.allclose(chip.potentials(), orc.potentials(), rtol=1e-9, atol=1e-12), t
    return chip, orc


def check_batched(S, arch, net, steps, timing="simple"):
    """Whole run on the device in one call, per-step records compared afterwards."""
    chip, orc = make(S, arch, net)
    tot = chip.run(steps, timing, record=True)
    recs = chip.step_totals(0, steps)
    acc = {}
    for t in range(steps):
        b = orc.step(timing)
        for ka, kb in INT_KEYS:
            assert recs[ka][t] == b[kb], (t, ka)
        for k in DBL_KEYS:
            assert recs[k][t] == pytest.approx(b[k], rel=REL, abs=1e-30), (t, k)
        assert np.array_equal(chip.step_fired(t), (orc.status() == 3).astype(np.uint8)), t
        for k, v in b.items():
            acc[k] = acc.get(k, 0) + v
    for ka, kb in INT_KEYS:
        assert tot[ka] == acc[kb]
    for k in DBL_KEYS:
        assert tot[k] == pytest.approx(acc[k], rel=REL)
    assert np.array_equal(chip.potentials(), orc.potentials())
    return chip, orc, tot


def test_example_chip_simple(S):
    check_stepwise(S, *nets.example(S), steps=30)


def test_example_chip_probe_values(S):
    """The survey-time outputs of the reference for C1 from the GPU path, detailed timing: counters, spike and
    potential traces are exact pins; the two 3-digit doubles (energy, sim_time) guard against gross errors only --
    `detailed` timestamps and energies are PARITY UNPINNED."""
    arch, net = nets.example(S)
    chip = S.SpikingChip(arch)
    chip.load(net)
    r = chip.sim(10, timing_model="detailed", spike_trace=True, potential_trace=True)
    assert (r["spikes"], r["packets_sent"], r["neurons_updated"], r["neurons_fired"]) == (5, 3, 20, 3)
    assert r["energy"]["total"] == pytest.approx(1.04e-09, rel=5e-3)
    assert r["sim_time"] == pytest.approx(1.24e-07, rel=5e-3)
    # out.1 fires at step 2 but group `out` does not set log_spikes (snn/example_snn.yaml:9-12)
    assert r["spike_trace"][:3] == [[("in", 1)], [], [("in", 1)]]
    assert r["potential_trace"][:4] == [[0.0, 0.0], [1.0, 0.0], [1.0, -4.0], [2.0, -1.0]]


def test_example_chip_detailed_messages(S):
    arch, net = nets.example(S)
    chip, orc = make(S, arch, net)
    for t in range(12):
        a = chip.run(1, "detailed", record=True, messages=True)
        b = orc.step("detailed")
        assert a["sim_time"] == b["sim_time"], t  # same serial algorithm on identical inputs: bit-exact
        ma, mb = chip.step_messages(0), orc.messages()
        assert len(ma) == len(mb)
        for name in ma.dtype.names:
            assert np.array_equal(ma[name], mb[name]), (t, name)


def test_random_loihi_before_soma(S):
    arch, net = nets.random_loihi(S, n_tiles=4, neurons_per_core=100, out_degree=24, arch_kind="loihi", refractory=True)
    check_stepwise(S, arch, net, steps=25)


def test_random_loihi_delay_line(S):
    arch, net = nets.random_loihi(S, n_tiles=4, neurons_per_core=100, out_degree=24, arch_kind="large", delays=True)
    check_stepwise(S, arch, net, steps=30)


def test_random_loihi_detailed_messages(S):
    arch, net = nets.random_loihi(S, n_tiles=6, neurons_per_core=50, out_degree=16, arch_kind="loihi", seed=3)
    chip, orc = make(S, arch, net)
    for t in range(8):
        a = chip.run(1, "detailed", record=True, messages=True)
        b = orc.step("detailed")
        ma, mb = chip.step_messages(0), orc.messages()
        assert len(ma) == len(mb) > 0
        for name in ma.dtype.names:
            assert np.array_equal(ma[name], mb[name]), (t, name)
        assert a["sim_time"] == b["sim_time"]


def test_detailed_scheduler_many_messages_in_flight(S):
    """Thousands of messages per step over 64 cores: the heap-based retirement of in-flight messages performs the
    reference's density / rolling-average updates in the reference's order (the oracle keeps its per-core scan)."""
    arch, net = nets.random_loihi(S, n_tiles=16, neurons_per_core=40, out_degree=120, arch_kind="loihi", p_fire=0.3, seed=21)
    chip, orc = make(S, arch, net)
    for t in range(5):
        a = chip.run(1, "detailed", record=True, messages=True)
        b = orc.step("detailed")
        ma, mb = chip.step_messages(0), orc.messages()
        assert len(ma) == len(mb) > 1000
        for name in ma.dtype.names:
            assert np.array_equal(ma[name], mb[name]), (t, name)
        assert a["sim_time"] == b["sim_time"]


def test_detailed_scheduler_threads_match_inline(S):
    """`scheduler_threads=n` (src/chip.cpp:291-349, src/schedule.cpp:182-206) overlaps the NoC schedule of finished
    steps with the GPU; every timestep's schedule is independent, so results equal the inline run bit for bit."""
    arch, net = nets.random_loihi(S, n_tiles=6, neurons_per_core=50, out_degree=16, arch_kind="loihi", seed=3)
    chip, orc = make(S, arch, net)
    chip2, _ = make(S, arch, net)
    a = chip.sim(40, timing_model="detailed", scheduler_threads=0, perf_trace=True, message_trace=True)
    b = chip2.sim(40, timing_model="detailed", scheduler_threads=4, perf_trace=True, message_trace=True)
    assert a["sim_time"] == b["sim_time"] and a["energy"] == b["energy"]
    assert a["perf_trace"] == b["perf_trace"]
    assert a["message_trace"] == b["message_trace"]
    total = 0.0
    for _ in range(40):
        total += orc.step("detailed")["sim_time"]
    assert a["sim_time"] == total


def _srand1():
    import ctypes
    ctypes.CDLL(None).srand(1)  # the oracle draws from the process-wide std::rand(), like the reference


def test_poisson_inputs_and_lif_noise_file(S, tmp_path):
    """Rows a22/a24 on the GPU: host-generated value streams (std::mt19937 Poisson draws, LIF noise file)."""
    arch, net = nets.stochastic(S, tmp_path)
    check_stepwise(S, arch, net, steps=150)


def test_truenorth_random_mask(S):
    """Row a23 on the GPU: `std::rand() & random_mask` in the TrueNorth threshold test."""
    _srand1()
    check_stepwise(S, *nets.stochastic_truenorth(S), steps=40)


def test_value_streams_batched_in_chunks(S, tmp_path, monkeypatch):
    """A batched run uploads the value streams in chunks; totals and spikes equal the step-by-step run."""
    monkeypatch.setenv("SANAFE_EXT_CHUNK_STEPS", "7")
    arch, net = nets.stochastic(S, tmp_path)
    chip, orc, tot = check_batched(S, arch, net, steps=45)
    assert tot["neurons_fired"] > 0


def _random_configuration(S, seed):
    """Randomly drawn shapes around the thresholds of the delivery kernel's paths: chunk boundaries (256 axons), the
    stream/gather switch (16 spiking axons per chunk), single- and multi-slice cores, every supported buffer position
    and dendrite, the three synapse formats."""
    rng = np.random.default_rng(seed)
    kind = ["large", "loihi", "before_dendrite", "loihi_delay"][int(rng.integers(0, 4))]
    delays = kind == "loihi_delay" or bool(kind == "large" and rng.integers(0, 2))
    npc = int(rng.choice([37, 64, 200, 257, 511, 700]))
    cores = int(rng.integers(2, 7))
    n = npc * cores
    return nets.random_loihi(S, n_tiles=2, neurons_per_core=npc, cores_used=cores,
                             out_degree=int(min(n, rng.choice([5, 40, 150, 400]))), arch_kind="loihi" if kind == "loihi_delay" else kind,
                             delays=delays, p_fire=float(rng.choice([0.01, 0.05, 0.3])), seed=seed,
                             weights=str(rng.choice(["int", "int", "int12", "float"] if not delays else ["int", "int12"])),
                             refractory=bool(rng.integers(0, 2)),
                             dendrite="loihi_dendrites_delay" if kind == "loihi_delay" else None)


def _fuzz_seeds():
    """16 seeds by default; SANAFE_FUZZ_SEEDS="lo:hi" widens the sweep for a one-off soak run."""
    spec = os.environ.get("SANAFE_FUZZ_SEEDS")
    if spec:
        lo, hi = (int(x) for x in spec.split(":"))
        return list(range(lo, hi))
    return list(range(11, 27))


@pytest.mark.parametrize("seed", _fuzz_seeds())
def test_random_configurations(S, seed, monkeypatch):
    if seed % 2:
        monkeypatch.setenv("SANAFE_MIN_SLICE_AXONS", "256")
        monkeypatch.setenv("SANAFE_TARGET_SLICES", "100000")
    arch, net = _random_configuration(S, seed)
    chip, orc = make(S, arch, net)
    exact = chip.device_layout()["syn_format"] not in (2, 4, 6)  # streamed non-integer weights: sums in arrival order
    tot = chip.run(14, "simple", record=True)
    recs = chip.step_totals(0, 14)
    for t in range(14):
        b = orc.step("simple")
        for ka, kb in INT_KEYS:
            assert recs[ka][t] == b[kb], (t, ka)
        for k in DBL_KEYS:
            assert recs[k][t] == pytest.approx(b[k], rel=REL, abs=1e-30), (t, k)
        assert np.array_equal(chip.step_fired(t), (orc.status() == 3).astype(np.uint8)), t
    if exact:
        assert np.array_equal(chip.potentials(), orc.potentials())
    else:
        assert np.allclose(chip.potentials(), orc.potentials(), rtol=1e-9, atol=1e-12)


@pytest.mark.parametrize("net_kind", ["truenorth_push_only", "truenorth", "loihi_sparse"])
def test_push_delivery_on_steps_with_few_spikes(S, monkeypatch, net_kind):
    """C4-like activity (a fraction of a percent of the neurons fire): the NEURON launch delivers the spikes itself -- the
    wavefront of a 64-slot chunk walks the out-synapse lists of its fired neurons and adds to the next step's row of the
    time-step buffer -- and the delivery launch, which probes every inbound axon, is not launched (push or pull decided per
    step by the host from the events of the step sixteen before, which the device publishes in a pinned ring), or never
    launched at all on chips with at most one synapse per neuron (push-only: TrueNorth's one edge per neuron).  Same spikes, potentials, counters, energies and sim_time
    as the oracle and as the pull path (SANAFE_PUSH=0)."""
    if net_kind.startswith("truenorth"):
        arch, net = nets.truenorth_net(S, n_tiles=32, neurons_per_core=256)
        monkeypatch.setenv("SANAFE_PUSH_MAX_EVENTS", "2000")  # 8,192 events in step 1
        if net_kind == "truenorth":
            monkeypatch.setenv("SANAFE_PUSH_ONLY", "0")  # decide per step: the first steps (every neuron fires in step 1) pull
    else:
        arch, net = nets.random_loihi(S, n_tiles=4, neurons_per_core=200, out_degree=30, arch_kind="loihi", p_fire=0.01, seed=43)
        monkeypatch.setenv("SANAFE_PUSH_MAX_EVENTS", "5000")  # (a chip this small would pull: one probe per axon is cheap)
    chip, orc, tot = check_batched(S, arch, net, steps=40)  # one sim() call: the host decides while the device runs
    lay = chip.device_layout()
    assert lay["push_enabled"] and lay["push_only"] == (net_kind == "truenorth_push_only"), lay
    # step t (1-based) is pushed when step (t - 16) rounded down to a multiple of 4 caused at most SANAFE_PUSH_MAX_EVENTS events:
    # a pure function of the run
    ev = chip.step_totals(0, 40)["spikes"]
    limit = int(os.environ["SANAFE_PUSH_MAX_EVENTS"])
    expect = 40 if net_kind == "truenorth_push_only" else sum(1 for t in range(1, 41) if (t - 16) // 4 * 4 >= 1 and ev[(t - 16) // 4 * 4 - 1] <= limit)
    assert lay["pushed_steps"] == expect and (expect >= 12 or net_kind == "truenorth"), (lay, expect)
    monkeypatch.setenv("SANAFE_PUSH", "0")
    pull = S.SpikingChip(arch)
    pull.load(net)
    assert not pull.device_layout()["push_enabled"]
    a = pull.run(40, "simple")
    for k in ("spikes", "packets_sent", "neurons_updated", "neurons_fired", "total_hops"):
        assert a[k] == tot[k], k
    for k in DBL_KEYS:
        assert a[k] == pytest.approx(tot[k], rel=1e-12, abs=1e-30), k
    assert np.array_equal(pull.potentials(), chip.potentials())


def test_taps_dendrites(S):
    """Row a21 on the GPU: `taps` dendrites -- per-tap charge through the delivery rows, the RC line advanced by
    taps_kernel, tap 0 handed to the soma through the time-step buffer."""
    arch, net = nets.taps_dendrites(S)
    chip, orc, tot = check_batched(S, arch, net, steps=45)
    assert tot["neurons_fired"] > 300
    chip.reset()
    orc.reset()
    for t in range(5):
        a, b = chip.run(1, "simple"), orc.step("simple")
        assert a["neurons_fired"] == b["neurons_fired"]
        assert np.array_equal(chip.potentials(), orc.potentials()), t


def test_neurons_sharing_an_input_unit(S):
    check_batched(S, *nets.shared_input_units(S), steps=60)


def test_plain_accumulator_inside_dendrite_quirk(S):
    arch, net = nets.random_loihi(S, n_tiles=2, neurons_per_core=64, out_degree=12, arch_kind="large")
    for g in net._order:
        g.dendrite_hw[:] = net.strings("loihi_dendrites")
    check_stepwise(S, arch, net, steps=12)


@pytest.mark.parametrize("small", [True, False])
def test_truenorth(S, monkeypatch, small):
    """256 axons per core = one chunk per delivery slice: 64-thread delivery workgroups (default) and the 256-thread ones."""
    if not small:
        monkeypatch.setenv("SANAFE_DELIVER_SMALL", "0")
    check_stepwise(S, *nets.truenorth_net(S, n_tiles=16, neurons_per_core=256), steps=20)


@pytest.mark.parametrize("force,fmt", [(None, 8), ("4", 4), ("2", 2)])
def test_float_weights(S, monkeypatch, force, fmt):
    """Non-integer weights.  Default: ordered delivery (format 8) -- every accumulator folds its events in the reference's
    order, so potentials are BIT-EQUAL to the oracle's.  Forced: the streaming layouts, 4 + 8 bytes per synapse (index-coded
    words, format 4) and the gather-only fall-back (format 2), whose atomics add in arrival order (1e-9); p_fire 0.5 makes
    whole chunks stream."""
    if force:
        monkeypatch.setenv("SANAFE_SYN_FORMAT", force)
    arch, net = nets.random_loihi(S, n_tiles=2, neurons_per_core=64, out_degree=48, arch_kind="loihi", weights="float", p_fire=0.5)
    chip, _ = check_stepwise(S, arch, net, steps=20, exact_v=(force is None))
    assert chip.device_layout()["syn_format"] == fmt


@pytest.mark.parametrize("kind", ["loihi", "large_delays", "dictionary", "no_dictionary_forced", "multi_slice", "global_bitmap"])
def test_ordered_delivery_is_bit_exact_over_200_steps(S, monkeypatch, kind):
    """VERDICT r2 #1b: with non-integer weights the HIP path adds a step's synaptic currents per accumulator in the
    reference's order (source core, source neuron, connection: src/chip.cpp:661-690, 748-761), continuing the value the
    delay line already holds (`value_or(0.0) + current`, src/models.cpp:96-131).  200 steps of spikes AND potentials equal
    the oracle's bit for bit; a second chip reproduces the run bit for bit."""
    delays = kind == "large_delays"
    arch, net = nets.random_loihi(S, n_tiles=3, neurons_per_core=96, out_degree=60, arch_kind="large" if delays else "loihi",
                                  delays=delays, weights="float", p_fire=0.25, seed=41)
    if kind in ("dictionary", "no_dictionary_forced"):  # 20 distinct non-integer weights: 4-byte entries with a 5-bit code
        rng = np.random.default_rng(6)
        table = rng.normal(size=20) * 3.0
        for blk in net._edge_blocks:
            blk[2][:] = table[rng.integers(0, 20, size=len(blk[2]))]
    if kind == "no_dictionary_forced":
        monkeypatch.setenv("SANAFE_ORDERED_NO_DICT", "1")
    if kind == "multi_slice":  # many delivery slices per core: only the processing-delay sums follow the slices here
        monkeypatch.setenv("SANAFE_MIN_SLICE_AXONS", "256")
        monkeypatch.setenv("SANAFE_TARGET_SLICES", "100000")
    if kind == "global_bitmap":
        monkeypatch.setenv("SANAFE_ORDERED_LDS_BITS", "0")  # probe the bitmap in global memory (chips beyond ~300 k neurons)
    chip, orc = make(S, arch, net)
    assert chip.device_layout()["syn_format"] == 8
    fired = []
    for t in range(200):
        a, b = chip.run(1, "simple"), orc.step("simple")
        for ka, kb in INT_KEYS:
            assert a[ka] == b[kb], (t, ka, a[ka], b[kb])
        for k in DBL_KEYS:
            assert a[k] == pytest.approx(b[k], rel=REL, abs=1e-30), (t, k)
        st = orc.status()
        assert np.array_equal(chip.status(), st), t
        assert np.array_equal(chip.potentials(), orc.potentials()), t  # bit for bit
        fired.append(int((st == 3).sum()))
    assert sum(fired[100:]) > 0
    chip2 = S.SpikingChip(arch)
    chip2.load(net)
    chip2.run(200, "simple")
    assert np.array_equal(chip2.potentials(), chip.potentials()) and np.array_equal(chip2.status(), chip.status())


@pytest.mark.parametrize("force,fmt", [(None, 3), ("1", 1)])
def test_twelve_bit_integer_weights(S, monkeypatch, force, fmt):
    """12-bit integer weights: 4 bytes per synapse, streamable (format 3) or gather-only (format 1)."""
    if force:
        monkeypatch.setenv("SANAFE_SYN_FORMAT", force)
    arch, net = nets.random_loihi(S, n_tiles=2, neurons_per_core=64, out_degree=48, arch_kind="loihi", weights="int12", p_fire=0.5)
    chip, _ = check_stepwise(S, arch, net, steps=20)
    assert chip.device_layout()["syn_format"] == fmt


@pytest.mark.parametrize("force,fmt", [(None, 7), ("6", 6), ("0", 0), ("1", 1)])
def test_small_integer_weights_four_layouts(S, monkeypatch, force, fmt):
    """Weights in {-8..8}: 2-byte dictionary-coded words with integer accumulators (format 7, the default) or fp64
    accumulators (format 6), 4-byte int8 words (format 0) and the gather-only 12-bit words (format 1) -- dense cores
    so that whole chunks stream, half of the neurons firing."""
    if force:
        monkeypatch.setenv("SANAFE_SYN_FORMAT", force)
    arch, net = nets.random_loihi(S, n_tiles=1, neurons_per_core=256, out_degree=180, arch_kind="loihi", p_fire=0.5, seed=29)
    chip, _ = check_stepwise(S, arch, net, steps=12)
    lay = chip.device_layout()
    assert lay["syn_format"] == fmt and lay["n_compact_slices"] > 0


@pytest.mark.parametrize("weights,force,fmt", [("int", "0", 0), ("int12", None, 3)])
@pytest.mark.parametrize("int_acc", [True, False])
def test_integer_and_fp64_accumulators_of_the_4_byte_words(S, monkeypatch, weights, force, fmt, int_acc):
    """Formats 0 and 3 sum integer weights in 32-bit integer LDS accumulators when the per-accumulator bounds hold
    (sanafe_hip_get_acc_shift > 0) and in fp64 otherwise (forced here with SANAFE_INT_ACC=0): the same results, bit for
    bit, on a streamed network and with synaptic delays."""
    if force:
        monkeypatch.setenv("SANAFE_SYN_FORMAT", force)
    if not int_acc:
        monkeypatch.setenv("SANAFE_INT_ACC", "0")
    kw = {} if weights == "int" else {"weights": weights}
    arch, net = nets.random_loihi(S, n_tiles=1, neurons_per_core=256, out_degree=180, arch_kind="loihi", p_fire=0.5, seed=41, **kw)
    chip, _ = check_stepwise(S, arch, net, steps=10)
    lay = chip.device_layout()
    assert lay["syn_format"] == fmt and (lay["acc_shift"] > 0) == int_acc and lay["n_compact_slices"] > 0
    arch, net = nets.random_loihi(S, n_tiles=2, neurons_per_core=128, out_degree=60, arch_kind="large", delays=True, p_fire=0.05,
                                  seed=42, **kw)
    chip, _ = check_stepwise(S, arch, net, steps=10)
    assert (chip.device_layout()["acc_shift"] > 0) == int_acc


@pytest.mark.parametrize("force", [None, "6", "0", "1"])
def test_last_event_wins_on_streamed_chunks(S, monkeypatch, force):
    """Buffer before the dendrite unit (only the LAST event of a step reaches the accumulator, src/chip.cpp:759) on dense
    cores: the stream path of every layout keeps the position of the last event in delivery order, not a sum."""
    if force:
        monkeypatch.setenv("SANAFE_SYN_FORMAT", force)
    arch, net = nets.random_loihi(S, n_tiles=1, neurons_per_core=256, out_degree=170, arch_kind="before_dendrite", p_fire=0.5, seed=43)
    chip, _ = check_stepwise(S, arch, net, steps=10)
    lay = chip.device_layout()
    assert lay["syn_format"] == {None: 7, "6": 6, "0": 0, "1": 1}[force] and lay["n_compact_slices"] > 0


@pytest.mark.parametrize("force", [None, "0"])
def test_several_runs_per_wavefront(S, monkeypatch, force):
    """One delivery slice per core with ~15 k inbound axons = 60 chunks: every wavefront streams two runs of 8 chunks,
    the last run of the slice is short and its last chunk partial."""
    monkeypatch.setenv("SANAFE_TARGET_SLICES", "16")
    monkeypatch.setenv("SANAFE_MIN_SLICE_AXONS", "100000")
    if force:
        monkeypatch.setenv("SANAFE_SYN_FORMAT", force)
    arch, net = nets.random_loihi(S, n_tiles=4, neurons_per_core=960, out_degree=64, arch_kind="large", p_fire=0.3, seed=47)
    chip, _ = check_stepwise(S, arch, net, steps=6)
    lay, info = chip.device_layout(), chip.info()
    assert lay["syn_format"] == (7 if force is None else 0) and lay["n_compact_slices"] == info["n_slices"] == 16
    assert info["n_axons"] > 16 * 9000


@pytest.mark.parametrize("force", [None, "6", "0"])
def test_axons_with_two_hundred_synapses(S, monkeypatch, force):
    """One core, every neuron sends all of its 200 synapses there: a single axon's words span 25 lanes of a group-row
    (formats 6, 7: only its first word carries the first-synapse bit) and several 16-byte groups."""
    if force:
        monkeypatch.setenv("SANAFE_SYN_FORMAT", force)
    arch, net = nets.random_loihi(S, n_tiles=1, neurons_per_core=256, cores_used=1, out_degree=200, arch_kind="loihi", p_fire=0.3, seed=53)
    chip, _ = check_stepwise(S, arch, net, steps=8)
    lay = chip.device_layout()
    assert lay["syn_format"] == {None: 7, "6": 6, "0": 0}[force] and lay["n_compact_slices"] > 0


def test_dictionary_coded_float_weights(S, monkeypatch):
    """Format 6 is a dictionary, not an integer format: 20 distinct non-integer weights code into it as well (forced: the
    default for non-integer weights is the ordered layout, format 8)."""
    monkeypatch.setenv("SANAFE_SYN_FORMAT", "6")
    arch, net = nets.random_loihi(S, n_tiles=1, neurons_per_core=256, out_degree=150, arch_kind="loihi", p_fire=0.4, seed=31)
    rng = np.random.default_rng(5)
    table = rng.normal(size=20) * 3.0
    for blk in net._edge_blocks:  # (src, dst, weight, delay, synapse_hw)
        blk[2][:] = table[rng.integers(0, 20, size=len(blk[2]))]
    chip, _ = check_stepwise(S, arch, net, steps=12, exact_v=False)
    assert chip.device_layout()["syn_format"] == 6


def test_integer_accumulators_zero_sums_and_sparse_spikes(S, monkeypatch):
    """Format 7 keeps "an event arrived" apart from "the sum is zero": weights +-1 cancel often, and the buffer still
    holds a value (src/chip.cpp:759).  Low firing rate: the gather path adds into the same integer accumulators.
    A dictionary too wide for the bounds of format 7 (|w| up to 2^20 on dense cores is fine; 2^21 is not integer-coded)
    falls back to the 4-byte / fp64-accumulator layouts."""
    arch, net = nets.random_loihi(S, n_tiles=1, neurons_per_core=256, out_degree=150, arch_kind="loihi", p_fire=0.4, seed=37)
    rng = np.random.default_rng(9)
    for blk in net._edge_blocks:
        blk[2][:] = rng.choice([-1.0, 1.0], size=len(blk[2]))
    chip, _ = check_stepwise(S, arch, net, steps=12)
    assert chip.device_layout()["syn_format"] == 7
    arch, net = nets.random_loihi(S, n_tiles=1, neurons_per_core=256, out_degree=150, arch_kind="loihi", p_fire=0.02, seed=38)
    chip, _ = check_stepwise(S, arch, net, steps=12)
    assert chip.device_layout()["syn_format"] == 7
    arch, net = nets.random_loihi(S, n_tiles=1, neurons_per_core=256, out_degree=150, arch_kind="loihi", p_fire=0.4, seed=39)
    for blk in net._edge_blocks:
        blk[2][:] = rng.choice([-2097152.0, 3.0, 1048576.0], size=len(blk[2]))
    chip, _ = check_stepwise(S, arch, net, steps=8)
    assert chip.device_layout()["syn_format"] == 6  # integers, but no int8/int12 form and outside the bounds of 7
    # one weight value (a dictionary of one), but 255 x 127 per accumulator needs a 17-bit biased weight: too wide for the
    # 16-bit table of format 7 -> 4-byte int8 words, still with integer accumulators (shift 16)
    arch, net = nets.random_loihi(S, n_tiles=1, neurons_per_core=256, cores_used=1, out_degree=255, arch_kind="loihi", p_fire=0.4, seed=40)
    for blk in net._edge_blocks:
        blk[2][:] = 127.0
    chip, _ = check_stepwise(S, arch, net, steps=8)
    lay = chip.device_layout()
    assert lay["syn_format"] == 0 and lay["acc_shift"] == 16


@pytest.mark.parametrize("weights,fmt", [("int12", 3), ("float", 4)])
def test_streamed_delay_lines_other_formats(S, monkeypatch, weights, fmt):
    """The stream path of formats 3 / 4 with synaptic delays (several accumulator rows) on dense cores."""
    if weights == "float":
        monkeypatch.setenv("SANAFE_SYN_FORMAT", "4")  # (the default for non-integer weights is the ordered layout)
    arch, net = nets.random_loihi(S, n_tiles=1, neurons_per_core=256, out_degree=200, arch_kind="large", delays=True,
                                  weights=weights, p_fire=0.4, seed=13)
    chip, _ = check_stepwise(S, arch, net, steps=10, exact_v=(weights != "float"))
    lay = chip.device_layout()
    assert lay["syn_format"] == fmt and lay["n_compact_slices"] > 0


def test_mixed_axon_record_modes(S):
    """Dense recurrent cores pack their inbound axons as 2-byte delta records; a slice holding an axon with
    more than 255 synapses keeps the 8-byte records.  Both forms in one chip, with synaptic delays."""
    arch, net = nets.random_loihi(S, n_tiles=1, neurons_per_core=512, out_degree=200, arch_kind="large", delays=True,
                                  p_fire=0.2, seed=8)
    g = net._order[0]
    rng = np.random.default_rng(3)
    posts = rng.choice(512, size=300, replace=False)
    pairs = np.stack([np.full(300, 700, dtype=np.int64), posts.astype(np.int64)], axis=1)  # neuron 700 -> 300 neurons of core 0
    g.connect_neurons_sparse(g, {"weight": np.ones(300), "delay": rng.integers(0, 6, size=300)}, pairs, narrow_float=False)
    chip, _ = check_stepwise(S, arch, net, steps=12)
    lay, info = chip.device_layout(), chip.info()
    assert lay["syn_format"] == 0  # 6 delay rows x 513 accumulators: too many for the 2-byte words -> 4-byte int8 words
    assert 0 < lay["n_compact_slices"] < info["n_slices"]


def test_multi_slice_core(S, monkeypatch):
    """A core whose inbound axon list is split over several delivery workgroups (atomic write-back)."""
    monkeypatch.setenv("SANAFE_TARGET_SLICES", "100000")
    monkeypatch.setenv("SANAFE_MIN_SLICE_AXONS", "512")
    arch, net = nets.random_loihi(S, n_tiles=1, neurons_per_core=1000, cores_used=4, out_degree=600, arch_kind="loihi",
                                  p_fire=0.3)
    chip, orc = check_stepwise(S, arch, net, steps=6)
    assert chip.info()["n_slices"] > 4
    assert chip.device_layout()["n_compact_slices"] > 4  # dense inbound axon lists: 2-byte delta records


def test_batched_run_records(S):
    arch, net = nets.random_loihi(S, n_tiles=4, neurons_per_core=128, out_degree=32, arch_kind="large", delays=True, seed=5)
    check_batched(S, arch, net, steps=40)


@pytest.mark.slow
def test_tutorial5_dvs_golden_on_gpu(S):
    """tutorial/tutorial_5_dvs.ipynb: neurons_fired == 365277 after 1000 steps -- from the HIP path."""
    arch, net = nets.tutorial5_dvs(S)
    chip = S.SpikingChip(arch)
    chip.load(net)
    r = chip.sim(1000, timing_model="simple")
    assert r["neurons_fired"] == 365277


@pytest.mark.slow
def test_dvs_yaml_c2_spike_trace(S):
    """Config C2 at its full length: loihi + dvs.yaml, 1000 timesteps through SpikingChip.sim() with the spike trace of
    all 18,678 neurons (log_spikes) and the potential trace of group 1 (log_potential, 3,600 neurons) -- the spike
    trace bit-exact against the oracle step by step, every per-step potential of group 1 equal (integer weights and
    thresholds: exact; BASELINE asks for <= 1e-6 relative)."""
    arch, net = nets.dvs_yaml(S)
    chip, orc = make(S, arch, net)
    steps = 1000
    r = chip.sim(steps, timing_model="simple", spike_trace=True, potential_trace=True, perf_trace=True)
    logged = chip._trace_order[chip._log_potential[chip._trace_order]]
    assert len(logged) == 3600 and len(r["spike_trace"]) == steps and len(r["potential_trace"]) == steps
    labels = chip._labels()
    order = chip._trace_order
    fired_total = 0
    for t in range(steps):
        b = orc.step("simple")
        st = orc.status()
        want = [labels[int(g)] for g in order[(st[order] == 3) & chip._log_spikes[order]]]
        assert r["spike_trace"][t] == want, t
        assert np.array_equal(np.asarray(r["potential_trace"][t]), orc.potentials()[logged]), t
        assert r["perf_trace"]["fired"][t] == b["neurons_fired"] and r["perf_trace"]["spikes"][t] == b["spike_count"], t
        assert r["perf_trace"]["total_energy"][t] == pytest.approx(b["total_energy"], rel=REL), t
        assert r["perf_trace"]["sim_time"][t] == pytest.approx(b["sim_time"], rel=REL), t
        fired_total += b["neurons_fired"]
    assert r["neurons_fired"] == fired_total > 10000


def test_set_bias_between_sims(S):
    """The DVS workflow rewrites input biases between sim() calls (scripts/tcad2025/dvs_gesture.py:140-151)."""
    arch, net = nets.random_loihi(S, n_tiles=2, neurons_per_core=64, out_degree=8, arch_kind="loihi", p_fire=0.0)
    chip, orc = make(S, arch, net)
    rng = np.random.default_rng(0)
    for frame in range(3):
        bias = rng.integers(0, 40, size=net.neuron_count).astype(np.float64)
        chip.set_bias("n", bias)
        for g in range(net.neuron_count):
            orc.set_neuron_attr(g, "bias", (2, float(bias[g]), None, None))
        for t in range(5):
            a, b = chip.run(1, "simple"), orc.step("simple")
            assert a["neurons_fired"] == b["neurons_fired"]
            assert np.array_equal(chip.potentials(), orc.potentials())


def test_set_attributes_between_sims(S):
    """MappedNeuron.set_attributes with any soma attribute (src/mapped.cpp:113-166): the neuron moves to the
    parameter class holding the new value; bias and potential are patched in place."""
    D = S.description
    arch, net = nets.random_loihi(S, n_tiles=2, neurons_per_core=64, out_degree=12, arch_kind="loihi", p_fire=0.15, seed=6)
    chip, orc = make(S, arch, net)
    rng = np.random.default_rng(1)
    n = net.neuron_count
    groups = chip.mapped_neuron_groups
    changes = [("threshold", lambda: int(rng.integers(20, 90))), ("reset", lambda: int(rng.integers(-5, 5))),
               ("leak_decay", lambda: float(rng.choice([0.5, 0.75, 0.9375]))), ("reset_mode", lambda: str(rng.choice(["soft", "hard"]))),
               ("refractory_delay", lambda: int(rng.integers(0, 3))), ("potential", lambda: float(rng.integers(-20, 60))),
               ("bias", lambda: int(rng.integers(0, 30))), ("reverse_threshold", lambda: float(-rng.integers(10, 40))),
               ("reverse_reset_mode", lambda: "saturate"), ("force_update", lambda: bool(rng.integers(0, 2)))]
    for round_ in range(4):
        for _ in range(40):
            g = int(rng.integers(0, n))
            key, gen = changes[int(rng.integers(0, len(changes)))]
            value = gen()
            groups["n"][g].set_attributes(model_attributes={key: value})
            orc.set_neuron_attr(g, key, D.py_to_attr(value))
        for t in range(6):
            a, b = chip.run(1, "simple"), orc.step("simple")
            for ka, kb in INT_KEYS:
                assert a[ka] == b[kb], (round_, t, ka)
            assert np.array_equal(chip.status(), orc.status()), (round_, t)
            assert np.array_equal(chip.potentials(), orc.potentials()), (round_, t)
    with pytest.raises(RuntimeError):
        groups["n"][0].set_attributes(model_attributes={"reset_mode": "sideways"})


def test_input_attributes_between_sims(S, tmp_path):
    """Scripts replace the `spikes` train of mapped input neurons before every sim() (scripts/computer2026/
    crossbar.py:301-309); `rate` and an existing Poisson probability can change too.  The train rewinds."""
    D = S.description
    arch, net = nets.stochastic(S, tmp_path)
    chip, orc = make(S, arch, net)
    rng = np.random.default_rng(5)
    gin = chip.mapped_neuron_groups["in"]
    base = dict(chip._built.groups)["in"][0]
    for frame in range(4):
        for i in range(len(gin)):
            train = [int(x) for x in rng.integers(0, 2, size=int(rng.integers(0, 12)))]
            gin[i].set_model_attributes(model_attributes={"spikes": train})
            orc.set_neuron_attr(base + i, "spikes", D.py_to_attr(train))
        i = int(rng.integers(0, len(gin)))
        for key, value in (("rate", float(rng.choice([0.0, 0.25, 0.5]))), ("poisson", float(rng.choice([0.0, 0.125, 0.5])))):
            gin[i].set_attributes(model_attributes={key: value})
            orc.set_neuron_attr(base + i, key, D.py_to_attr(value))
        for t in range(9):
            a, b = chip.run(1, "simple"), orc.step("simple")
            for ka, kb in INT_KEYS:
                assert a[ka] == b[kb], (frame, t, ka)
            assert np.array_equal(chip.status(), orc.status()), (frame, t)
            assert np.array_equal(chip.potentials(), orc.potentials()), (frame, t)


def test_value_stream_columns_that_come_and_go_after_load(S, tmp_path):
    """VERDICT r3 missing #5 (src/mapped.cpp:113-166, src/models.cpp:832-853, 752-758): set_attributes after load() may
    give an input neuron that had none a Poisson rate -- its unit's std::mt19937 has drawn at every update so far whatever
    the rate, so the new stream starts that many draws in -- and may give or take a TrueNorth neuron's `random_mask`, which
    changes which neurons draw from the process's std::rand() sequence.  The chip re-creates its device tables with the new
    value-stream columns and moves the run-time state over; every step against the oracle."""
    D = S.description
    arch, net = nets.stochastic(S, tmp_path, silent_inputs=(0, 3, 5))
    chip, orc = make(S, arch, net)
    gin = chip.mapped_neuron_groups["in"]
    base = dict(chip._built.groups)["in"][0]

    def steps(k, where):
        for t in range(k):
            a, b = chip.run(1, "simple"), orc.step("simple")
            for ka, kb in INT_KEYS:
                assert a[ka] == b[kb], (where, t, ka)
            assert np.array_equal(chip.status(), orc.status()), (where, t)
            assert np.array_equal(chip.potentials(), orc.potentials()), (where, t)

    steps(7, "before")
    for i, p in ((0, 0.5), (5, 0.25)):
        gin[i].set_attributes(model_attributes={"poisson": p})
        orc.set_neuron_attr(base + i, "poisson", D.py_to_attr(p))
    steps(9, "two new streams")
    gin[3].set_attributes(model_attributes={"poisson": 0.375})
    orc.set_neuron_attr(base + 3, "poisson", D.py_to_attr(0.375))
    steps(9, "a third")

    _srand1()
    arch, net = nets.stochastic_truenorth(S)
    chip, orc = make(S, arch, net)
    name = net._order[0].name
    group = chip.mapped_neuron_groups[name]
    rng = np.random.default_rng(3)
    steps(5, "truenorth before")
    for round_ in range(3):
        for g in rng.choice(len(group), size=25, replace=False):
            mask = int(rng.choice([0, 0, 1, 7, 63]))
            group[int(g)].set_attributes(model_attributes={"random_mask": mask})
            orc.set_neuron_attr(int(g), "random_mask", D.py_to_attr(mask))
        steps(6, ("truenorth", round_))


def test_reset(S):
    arch, net = nets.random_loihi(S, n_tiles=2, neurons_per_core=64, out_degree=8, arch_kind="loihi")
    chip, orc = make(S, arch, net)
    for t in range(5):
        chip.run(1, "simple")
        orc.step("simple")
    chip.reset()
    orc.reset()
    for t in range(5):
        a, b = chip.run(1, "simple"), orc.step("simple")
        assert a["neurons_fired"] == b["neurons_fired"]
        assert np.array_equal(chip.potentials(), orc.potentials())


def test_no_device_fallback_is_an_error(S):
    """A chip mapped without a device cannot simulate: there is no CPU execution path."""
    import ctypes as C
    arch, net = nets.example(S)
    L = S.chip.lib()
    built = S.to_desc(arch, net)
    h = C.c_void_p()
    assert L.sanafe_chip_create(C.addressof(built.desc), -1, 1, 0, C.byref(h)) == 0
    t = S.chip.Totals()
    assert L.sanafe_chip_sim(h, 1, 0, 0, C.byref(t)) != 0
    assert b"no CPU execution path" in L.sanafe_last_error()
    L.sanafe_chip_destroy(h)


@pytest.mark.parametrize("timing", ["simple", "detailed"])
def test_hodgkin_huxley_plugin_c5(S, timing):
    """Config C5: plugin somas loaded through `create_hodgkin_huxley` and evaluated by the host between the
    neuron and the delivery kernels; V within 1e-6 relative (same libm here: bit-exact), spike steps exact."""
    arch, net = nets.hodgkin_huxley(S)
    chip, orc = make(S, arch, net)
    fired_any = 0
    for t in range(120):
        a = chip.run(1, timing, record=True)
        b = orc.step(timing)
        for ka, kb in INT_KEYS:
            assert a[ka] == b[kb], (t, ka, a[ka], b[kb])
        for k in DBL_KEYS:
            assert a[k] == pytest.approx(b[k], rel=REL, abs=1e-30), (t, k)
        assert np.array_equal(chip.status(), orc.status()), t
        assert np.allclose(chip.potentials(), orc.potentials(), rtol=1e-6, atol=0), t
        fired_any += a["neurons_fired"]
    assert fired_any > 0


@pytest.mark.parametrize("timing", ["simple", "detailed"])
@pytest.mark.parametrize("position", ["inside", "before"])
def test_plugin_somas_pay_the_dendrite_of_the_neuron_pipeline(S, timing, position):
    """With the buffer before or inside the dendrite unit the neuron-processing pipeline runs dendrite -> soma
    (src/pipeline.cpp:268-310), so every update of a PLUGIN soma also costs the dendrite's energy_update /
    latency_update (energy, simple sim_time, detailed generation delays)."""
    D = S.description
    arch, net = nets.hodgkin_huxley(S)
    seen = set()
    for core in arch.cores():
        core.buffer_position = D.BUF_INSIDE_DENDRITE if position == "inside" else D.BUF_BEFORE_DENDRITE
        for u in core.units:
            if id(u) not in seen and (u.implements & D.IMPL_DENDRITE):
                seen.add(id(u))
                u.attributes["energy_update"] = (D.ATTR_DOUBLE, 1.5e-12, None, None)
                u.attributes["latency_update"] = (D.ATTR_DOUBLE, 2.0e-9, None, None)
    chip, orc = make(S, arch, net)
    e_dend = 0.0
    for t in range(40):
        a = chip.run(1, timing, record=True, messages=(timing == "detailed"))
        b = orc.step(timing)
        for ka, kb in INT_KEYS:
            assert a[ka] == b[kb], (t, ka, a[ka], b[kb])
        for k in DBL_KEYS:
            assert a[k] == pytest.approx(b[k], rel=REL, abs=1e-30), (t, k)
        assert np.array_equal(chip.status(), orc.status()), t
        if timing == "detailed":
            ma, mb = chip.step_messages(0), orc.messages()
            assert len(ma) == len(mb)
            for name in ("generation_delay", "sent_timestamp", "processed_timestamp"):
                assert np.array_equal(ma[name], mb[name]), (t, name)
        e_dend += a["dendrite_energy"]
    assert e_dend > 0.0


def test_cpp_frontend_yaml_on_gpu(S):
    """The product front-end end to end: C++ YAML reader -> C++ description -> mapper -> HIP, against the oracle
    fed by the independent Python twin (PyYAML) reading the same files."""
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    arch = S.load_arch(os.path.join(g, "mini_arch.yaml"))
    net = S.load_net(os.path.join(g, "mini_snn.yaml"), arch)
    chip = S.SpikingChip(arch)
    chip.load(net)
    t_arch = S.yaml_io.load_arch(os.path.join(g, "mini_arch.yaml"))
    t_net = S.yaml_io.load_net(os.path.join(g, "mini_snn.yaml"), t_arch)
    orc = OracleChip(S.to_desc(t_arch, t_net))
    fired = 0
    for t in range(60):
        a, b = chip.run(1, "detailed", record=True, messages=True), orc.step("detailed")
        for ka, kb in INT_KEYS:
            assert a[ka] == b[kb], (t, ka)
        for k in DBL_KEYS:
            assert a[k] == pytest.approx(b[k], rel=REL, abs=1e-30), (t, k)
        assert np.array_equal(chip.status(), orc.status()), t
        assert np.array_equal(chip.potentials(), orc.potentials()), t
        ma, mb = chip.step_messages(0), orc.messages()
        for name in ma.dtype.names:
            assert np.array_equal(ma[name], mb[name]), (t, name)
        fired += a["neurons_fired"]
    assert fired > 20


def test_cpp_frontend_sim_result_dict(S):
    """`sim()` result keys and trace shapes of the reference's Python API (src/pymodule.cpp:268-288, 692-705)."""
    import os
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    arch = S.load_arch(os.path.join(g, "mini_arch.yaml"))
    net = S.load_net(os.path.join(g, "mini_snn.yaml"), arch)
    chip = S.SpikingChip(arch)
    chip.load(net)
    r = chip.sim(10, timing_model="simple", spike_trace=True, potential_trace=True, perf_trace=True)
    assert set(r) == {"timestep_start", "timesteps_executed", "energy", "sim_time", "spikes", "packets_sent", "neurons_updated",
                      "neurons_fired", "spike_trace", "potential_trace", "neuron_trace", "perf_trace", "message_trace"}
    assert set(r["energy"]) == {"total", "synapse", "dendrite", "soma", "network"}
    assert (r["timestep_start"], r["timesteps_executed"]) == (1, 10)
    assert len(r["spike_trace"]) == 10 and len(r["potential_trace"]) == 10 and len(r["perf_trace"]["fired"]) == 10
    assert len(r["potential_trace"][0]) == 16 + 3  # grid + out log potentials
    r2 = chip.sim(5)  # cumulative: continues at step 11 with the default detailed model
    assert r2["timestep_start"] == 11 and r2["sim_time"] > 0
    assert chip.get_power() > 0


@pytest.mark.parametrize("which", ["truenorth", "loihi_delays", "loihi_refractory"])
def test_uniform_and_table_driven_neuron_kernels_agree(S, monkeypatch, which):
    """Chips whose neurons all carry one class word run neuron_kernel<MODEL, UNI=true> (parameters in scalar registers,
    no class word loaded); SANAFE_NEURON_GENERIC=1 forces the table-driven kernel on the same chip.  Both against the
    oracle, and bit-identical to each other."""
    if which == "truenorth":
        arch, net = nets.truenorth_net(S, n_tiles=16)
    elif which == "loihi_delays":
        arch, net = nets.random_loihi(S, n_tiles=4, neurons_per_core=100, out_degree=24, arch_kind="large", delays=True, seed=17)
    else:
        arch, net = nets.random_loihi(S, n_tiles=4, neurons_per_core=128, out_degree=24, arch_kind="loihi", refractory=True, seed=18)
    # every neuron the same class word?  (bias is per-slot data, not part of the class)
    chip_u, orc = check_stepwise(S, arch, net, steps=15)
    monkeypatch.setenv("SANAFE_NEURON_GENERIC", "1")
    chip_g = S.SpikingChip(arch)
    chip_g.load(net)
    a = chip_g.run(15, "simple", record=True)
    chip_u2 = None
    monkeypatch.delenv("SANAFE_NEURON_GENERIC")
    chip_u2 = S.SpikingChip(arch)
    chip_u2.load(net)
    b = chip_u2.run(15, "simple", record=True)
    # counters identical; energies and sim_time to the last bits only: the uniform kernel leaves COUNTS per wavefront and level 1
    # of the step reduction prices them once per core, the table-driven kernel prices per wavefront (different associations)
    for k in a:
        assert a[k] == (b[k] if isinstance(a[k], int) else pytest.approx(b[k], rel=1e-12, abs=1e-30)), k
    ra, rb = chip_g.step_totals(0, 15), chip_u2.step_totals(0, 15)
    for k in ra.dtype.names:
        assert np.array_equal(ra[k], rb[k]) if ra[k].dtype.kind == "i" else np.allclose(ra[k], rb[k], rtol=1e-12, atol=0), k
    assert np.array_equal(chip_g.potentials(), chip_u2.potentials())
    assert np.array_equal(chip_g.potentials(), orc.potentials())


@pytest.mark.parametrize("delays", [False, True])
def test_delay_line_behind_buffer_before_dendrite(S, delays):
    """`accumulator_with_delay` with `buffer_position: dendrite` (outside the unit): the message pipeline stops after the
    synapse, the kernel's buffer keeps the LAST event's current, and the unit's neuron-side call -- every step, without
    a synapse address -- integrates it with the delay of the unit's synapse address 0 (src/models.cpp:96-131,
    src/pipeline.hpp:460-508)."""
    arch, net = nets.random_loihi(S, n_tiles=2, neurons_per_core=80, out_degree=20, arch_kind="before_dendrite",
                                  dendrite="loihi_dendrites_delay", delays=delays, p_fire=0.2, seed=23)
    chip, orc = check_stepwise(S, arch, net, steps=30)
    assert chip.run(0, "simple")["neurons_fired"] == 0
