"""Event-driven delivery (event_deliver_kernel, DevImage::ev_*) against the oracle.

The reference walks only the synapses behind the messages that arrived (src/chip.cpp:738-764).  On chips too large for
push tables the device keeps a second, source-neuron-major copy of the format-7 synapse words; a step with few spikes
is delivered from it -- spike bitmap -> fired neurons -> their blocks of words -> LDS integer accumulators -- and a
busy step by the streaming kernel, decided per step on the device.  Both must give the oracle's spikes, potentials,
counters, energies and sim_time."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nets  # noqa: E402
from oracle.oracle import OracleChip  # noqa: E402
from test_gpu_parity import DBL_KEYS, INT_KEYS, check_batched, check_stepwise  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("p_fire,segments,group_cores,lpb,table", [(0.02, "1", "16", "4", "n"), (0.02, "3", "3", "8", "g"), (0.5, "2", "5", "4", "g"),
                                                                  (0.5, "2", "5", "4", "n")])
def test_every_step_by_events_matches_the_oracle(S, monkeypatch, p_fire, segments, group_cores, lpb, table):
    """SANAFE_EVENT=2: every step goes through the event kernel, the streaming kernel is never launched.  One and several
    segments of the source space (the last one empty: 4 tiles over 3 segments of 2), core groups that do not fill the
    8-way block -> group mapping, both lane shapes, sparse and dense activity (blocks longer than one batch slot), the
    neuron-major and the group-major copy of the block table (forced modes take the neuron-major one unless
    SANAFE_EVENT_SPARSE_EVENTS=0)."""
    monkeypatch.setenv("SANAFE_EVENT", "2")
    if table == "g":
        monkeypatch.setenv("SANAFE_EVENT_SPARSE_EVENTS", "0")
    monkeypatch.setenv("SANAFE_EVENT_SEGMENTS", segments)
    monkeypatch.setenv("SANAFE_EVENT_GROUP_CORES", group_cores)
    monkeypatch.setenv("SANAFE_EVENT_LPB", lpb)
    arch, net = nets.random_loihi(S, n_tiles=4, neurons_per_core=200, out_degree=90 if p_fire > 0.1 else 30, arch_kind="loihi",
                                  p_fire=p_fire, seed=51)
    chip, _ = check_stepwise(S, arch, net, steps=14)
    lay = chip.device_layout()
    ev = lay["event_layout"]
    assert lay["syn_format"] == 7 and ev is not None and ev["always"], lay
    assert ev["segments"] == int(segments) and ev["lanes_per_block"] == int(lpb), ev
    assert ev["groups"] == {"16": 2, "3": 6, "5": 4}[group_cores], ev  # (at most 15 cores of 256 slots fit 4,096 - 64 accumulators)
    assert lay["pushed_steps"] == 14, lay  # steps delivered by events are counted like pushed ones
    assert ev["sparse_steps"] == (14 if table == "n" else 0), ev


def test_event_or_stream_decided_per_step(S, monkeypatch):
    """SANAFE_EVENT=1: decided per step (few events sixteen steps earlier -> events) by the host, from the event counts the
    device publishes, while the device runs ahead: one sim() call.  A network whose activity grows, so both kernels run.
    Same result as the oracle and as the chip without the event layout."""
    monkeypatch.setenv("SANAFE_EVENT", "1")
    monkeypatch.setenv("SANAFE_EVENT_MAX_EVENTS", "7400")  # 4,530 events in step 1, 7,000-8,100 from step 9 on
    monkeypatch.setenv("SANAFE_EVENT_SPARSE_EVENTS", "7100")
    arch, net = nets.random_loihi(S, n_tiles=4, neurons_per_core=200, out_degree=30, arch_kind="loihi", p_fire=0.05, seed=52)
    chip, orc, tot = check_batched(S, arch, net, steps=40)
    lay = chip.device_layout()
    assert lay["event_layout"] is not None and not lay["event_layout"]["always"], lay
    # step t (1-based) goes by events when step (t - 16) rounded down to a multiple of 4 caused at most SANAFE_EVENT_MAX_EVENTS
    # events: the first nineteen and the busy ones stream
    ev = chip.step_totals(0, 40)["spikes"]
    expect = sum(1 for t in range(1, 41) if (t - 16) // 4 * 4 >= 1 and ev[(t - 16) // 4 * 4 - 1] <= 7400)
    assert lay["pushed_steps"] == expect and 4 <= expect <= 20, (lay, expect)
    # ... and with the neuron-major block table when it caused at most SANAFE_EVENT_SPARSE_EVENTS
    sparse = sum(1 for t in range(1, 41) if (t - 16) // 4 * 4 >= 1 and ev[(t - 16) // 4 * 4 - 1] <= 7100)
    assert lay["event_layout"]["sparse_steps"] == sparse and 0 < sparse < expect, (lay, sparse, expect)
    monkeypatch.setenv("SANAFE_EVENT", "0")
    monkeypatch.setenv("SANAFE_PUSH", "0")
    plain = S.SpikingChip(arch)
    plain.load(net)
    assert plain.device_layout()["event_layout"] is None
    a = plain.run(40, "simple")
    for k in ("spikes", "packets_sent", "neurons_updated", "neurons_fired", "total_hops"):
        assert a[k] == tot[k], k
    for k in DBL_KEYS:
        assert a[k] == pytest.approx(tot[k], rel=1e-12, abs=1e-30), k
    assert np.array_equal(plain.potentials(), chip.potentials())


@pytest.mark.slow
@pytest.mark.parametrize("p_fire,one_segment", [(0.01, False), (0.3, False), (0.01, True)])
def test_c3_delivery_shape_by_events(S, monkeypatch, p_fire, one_segment):
    """SURVEY 8(d)'s shape of C3 -- 262,144 source neurons on 1,024 cores of 256, 16 destination cores that hear from
    (nearly) every neuron -- delivered by events at p_fire 0.01 and at the headline's 0.3: ten steps
    against the oracle.  (The 16 destination cores lie in 16 different core groups: blocks of ~2.6 words, far below what the
    layout is built for by default -- SANAFE_EVENT=2 forces it; the table walk, the masks and the bounds are the same.)"""
    arch, net = nets.c3_delivery_shape(S, cores=1024, neurons_per_core=256, dest_cores=16, out_degree=41, p_fire=p_fire, delays=False)
    orc = OracleChip(S.to_desc(arch, net))
    monkeypatch.setenv("SANAFE_MIN_SLICE_AXONS", "16384")
    monkeypatch.setenv("SANAFE_EVENT", "2")
    if one_segment:
        # one segment of 256 tiles on 4 wavefronts: a wavefront's tiles span more than the 64 its 16-bit list entries can
        # address from the list's first tile -- the list is drained on the way (what a rank of an 8-GPU chip, 2,048 tiles, does)
        monkeypatch.setenv("SANAFE_EVENT_SEGMENTS", "1")
        monkeypatch.setenv("SANAFE_EVENT_WAVES", "4")
    chip = S.SpikingChip(arch)
    chip.load(net)
    lay, info = chip.device_layout(), chip.info()
    ev = lay["event_layout"]
    assert lay["syn_format"] == 7 and lay["n_bitmap_slices"] == info["n_slices"] and ev is not None and ev["always"], lay
    assert ev["groups"] == 69 and ev["segments"] == (1 if one_segment else 7) and ev["code_bits"] == 4, ev  # 15 cores of 256 per group (7 segments: 483 workgroups on 512 slots), 16 weight values
    fired = 0
    for t in range(10):
        a, b = chip.run(1, "simple", record=True), orc.step("simple")
        for ka, kb in INT_KEYS:
            assert a[ka] == b[kb], (t, ka, a[ka], b[kb])
        for k in DBL_KEYS:
            assert a[k] == pytest.approx(b[k], rel=1e-9, abs=1e-30), (t, k)
        st = orc.status()
        assert np.array_equal(chip.status(), st), t
        assert np.array_equal(chip.potentials(), orc.potentials()), t
        fired += int((st == 3).sum())
    assert fired > 0.8 * p_fire * 10 * 262144


@pytest.mark.parametrize("kind", ["push", "event"])
def test_split_steps_with_flushing_calls_between_the_halves(S, monkeypatch, kind):
    """sanafe_hip_step_neurons, then a call that flushes the pending reductions (synchronize, device_layout,
    read_totals), then sanafe_hip_step_deliver.  The two halves and level 1 of the step's reduction must agree on push or
    pull -- with a decision word on the device a flush between the halves cleared it, a pushed step was delivered twice and
    its per-core counters leaked into a later step.  The mode is now decided once per step on the host (launch_neurons) and
    travels as a kernel argument.  Against the oracle."""
    if kind == "push":
        arch, net = nets.random_loihi(S, n_tiles=4, neurons_per_core=200, out_degree=30, arch_kind="loihi", p_fire=0.01, seed=43)
        monkeypatch.setenv("SANAFE_PUSH_MAX_EVENTS", "5000")
    else:
        arch, net = nets.random_loihi(S, n_tiles=4, neurons_per_core=200, out_degree=30, arch_kind="loihi", p_fire=0.02, seed=44)
        monkeypatch.setenv("SANAFE_EVENT", "1")
        monkeypatch.setenv("SANAFE_EVENT_MAX_EVENTS", "1000000")
    chip = S.SpikingChip(arch)
    chip.load(net)
    orc = OracleChip(S.to_desc(arch, net))
    lay = chip.device_layout()
    assert lay["push_enabled"] and not lay["push_only"], lay
    assert (lay["event_layout"] is not None) == (kind == "event")
    chip.run(6, "simple")
    for _ in range(6):
        orc.step("simple")
    before, pushed_before = chip.read_totals(), chip.device_layout()["pushed_steps"]
    assert pushed_before == 0  # (no history yet: the first nineteen steps pull)
    ref = {k: 0 for _, k in INT_KEYS}
    ref_d = {k: 0.0 for k in DBL_KEYS}
    for t in range(16):
        chip.step_neurons()
        if t % 2 == 1:
            (chip.synchronize, chip.device_layout, chip.read_totals)[(t // 2) % 3]()
        chip.step_deliver("simple")
        b = orc.step("simple")
        for _, kb in INT_KEYS:
            ref[kb] += b[kb]
        for k in DBL_KEYS:
            ref_d[k] += b[k]
    assert np.array_equal(chip.status(), orc.status())
    assert np.array_equal(chip.potentials(), orc.potentials())
    after = chip.read_totals()
    for ka, kb in INT_KEYS:
        assert after[ka] - before[ka] == ref[kb], ka
    for k in DBL_KEYS:
        assert after[k] - before[k] == pytest.approx(ref_d[k], rel=1e-9, abs=1e-30), k
    assert chip.device_layout()["pushed_steps"] == 3  # steps 20 .. 22: decided from steps 4 .. 4 (16 back, rounded down to a multiple of 4)


def test_reset_and_state_carry_on_an_event_chip(S, monkeypatch):
    """A step delivered by events leaves the next step's input in the partial rows, not in the time-step buffer: `reset()`
    must drop it like the buffer rows, and `load(net, overwrite=False)` after timesteps -- which exports the chip's state
    (sanafe_hip_export_state) -- must fold it into the buffer first (event_fold_kernel).  Every step by events."""
    monkeypatch.setenv("SANAFE_EVENT", "2")

    def build(name, n, core_idx, seed, arch):
        rng = np.random.default_rng(seed)
        net = S.Network(name)
        g = net.create_neuron_group(name, n, {"reset": 0, "leak": 1}, "core_synapses", "core_dendrites", False, True, "core_soma")
        g.set_attribute_column("threshold", rng.integers(5, 30, size=n).astype(np.float64), integer=True)
        g.set_attribute_column("bias", np.where(rng.random(n) < 0.4, rng.integers(2, 7, size=n), 0).astype(np.float64), integer=True)
        src = np.repeat(np.arange(n, dtype=np.int64), 24)
        dst = rng.integers(0, n, size=24 * n).astype(np.int64)
        net.add_edges(src, dst, rng.integers(1, 5, size=24 * n).astype(np.float64), "core_synapses")
        cores = arch.cores()
        half = n // 2
        g.map_to_core(cores[core_idx], 0, half)
        g.map_to_core(cores[core_idx + 1], half, n)
        return net

    arch = S.presets.truenorth(n_tiles=8, width=4, height=2)
    chip = S.SpikingChip(arch)
    chip.load(build("a", 400, 0, 1, arch))
    assert chip.device_layout()["event_layout"] is not None and chip.device_layout()["event_layout"]["always"]
    first = chip.sim(12, timing_model="simple")
    # reset: a fresh run of the same chip equals the first one (pending input of step 12 is gone, potentials are zero)
    chip.reset()
    again = chip.sim(12, timing_model="simple")
    assert again["neurons_fired"] == first["neurons_fired"] > 0 and again["spikes"] == first["spikes"]
    # carry: add a second network after 12 more steps; A must continue exactly as it does alone
    chip.load(build("b", 300, 3, 2, arch))  # overwrite=False
    assert chip.n_neurons == 700 and chip.device_layout()["event_layout"] is not None
    second = chip.sim(13, timing_model="simple")
    alone_a, alone_b = S.SpikingChip(arch), S.SpikingChip(arch)
    alone_a.load(build("a", 400, 0, 1, arch))
    alone_b.load(build("b", 300, 3, 2, arch))
    alone_a.sim(12, timing_model="simple")
    alone_a.reset()
    alone_a.sim(12, timing_model="simple")
    a2, b2 = alone_a.sim(13, timing_model="simple"), alone_b.sim(13, timing_model="simple")
    assert second["neurons_fired"] == a2["neurons_fired"] + b2["neurons_fired"] and b2["neurons_fired"] > 0
    assert second["spikes"] == a2["spikes"] + b2["spikes"]
    v = chip.potentials()
    assert np.array_equal(v[:400], alone_a.potentials()) and np.array_equal(v[400:], alone_b.potentials())


def test_event_chip_under_detailed_timing_with_message_trace(S, monkeypatch):
    """`detailed` timing on a chip whose steps are delivered by events: the host rebuilds the messages from the per-step
    status log and schedules them (src/schedule.cpp:208-620) -- nothing of that depends on which kernel delivered the step.
    Totals, statuses and the message records against the oracle."""
    monkeypatch.setenv("SANAFE_EVENT", "2")
    arch, net = nets.random_loihi(S, n_tiles=4, neurons_per_core=200, out_degree=30, arch_kind="loihi", p_fire=0.03, seed=53)
    chip, orc, tot = check_batched(S, arch, net, steps=12, timing="detailed")
    assert chip.device_layout()["event_layout"] is not None and chip.device_layout()["pushed_steps"] == 12
    chip2 = S.SpikingChip(arch)
    chip2.load(net)
    orc2 = OracleChip(S.to_desc(arch, net))
    n_msgs = 0
    for t in range(6):
        a = chip2.run(1, "detailed", record=True, messages=True)
        b = orc2.step("detailed")
        assert a["sim_time"] == b["sim_time"], t  # same serial algorithm on identical inputs: bit-exact
        ma, mb = chip2.step_messages(0), orc2.messages()
        assert len(ma) == len(mb), t
        for name in ma.dtype.names:
            assert np.array_equal(ma[name], mb[name]), (t, name)
        n_msgs += len(ma)
    assert n_msgs > 0
