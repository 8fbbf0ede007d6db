"""Cores that cannot run on the device by construction -- rows a5 (buffer positions `soma`/inside and `axon_out`:
src/pipeline.cpp:268-310, src/mapped.cpp:27-58, 168-188) and the synapse / dendrite half of the plugin surface
(src/core.cpp:196-231, src/pipeline.hpp:69-301).

Their soma (or a plugin unit) is called once per synaptic EVENT, in delivery order, so the host library replays such
cores' neuron and message pipelines per timestep from the chip's spike bitmap (host/host_cores.cpp) while the device
runs every other core; statuses, spikes, counters, energies and simulated time must equal the oracle's, which runs
all five buffer positions generically."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nets  # noqa: E402
from oracle.oracle import OracleChip  # noqa: E402

pytestmark = pytest.mark.gpu

INT_KEYS = (("spikes", "spike_count"), ("packets_sent", "packets_sent"), ("neurons_updated", "neurons_updated"),
            ("neurons_fired", "neurons_fired"), ("total_hops", "total_hops"))
DBL_KEYS = ("total_energy", "synapse_energy", "dendrite_energy", "soma_energy", "network_energy", "sim_time")


def _compare(chip, orc, steps):
    fired = 0
    for t in range(steps):
        a, b = chip.run(1, "simple", record=True), orc.step("simple")
        for ka, kb in INT_KEYS:
            assert a[ka] == b[kb], (t, ka, a[ka], b[kb])
        for k in DBL_KEYS:
            assert a[k] == pytest.approx(b[k], rel=1e-9, abs=1e-30), (t, k, a[k], b[k])
        st = orc.status()
        assert np.array_equal(chip.status(), st), t
        assert np.array_equal(chip.potentials(), orc.potentials()), t
        assert np.array_equal(chip.step_fired(0), (st == 3).astype(np.uint8)), t
        fired += int((st == 3).sum())
    return fired


@pytest.mark.parametrize("where", ["device", "host"])
@pytest.mark.parametrize("position", ["soma_inside", "axon_out"])
def test_soma_inside_the_message_pipeline(S, monkeypatch, position, where):
    """Buffer positions 3 and 4 with TrueNorth somas: the soma updates once per synaptic event (and, inside the unit,
    once more in the neuron loop); behind the buffer before axon_out a neuron's status persists until the next event.
    With built-in units (`current_based`, `accumulator`, `truenorth`) such cores run ON THE DEVICE since round 4
    (msgsoma_kernel: one lane per post-synaptic neuron walks its inbound synapses in delivery order);
    SANAFE_HOST_CORES=1 keeps the host replay of round 3, which plugin units and other models still use."""
    if where == "host":
        monkeypatch.setenv("SANAFE_HOST_CORES", "1")
    arch, net = nets.host_cores(S, position=position)
    chip = S.SpikingChip(arch)
    chip.load(net)
    assert chip.device_layout()["msg_cores_on_device"] == (2 if where == "device" else 0)
    orc = OracleChip(S.to_desc(arch, net))
    fired = _compare(chip, orc, 40)
    host = chip.status().reshape(6, -1)[[2, 4]]
    assert fired > 200 and (host == 3).sum() + (host == 2).sum() > 0  # the host cores take part
    # reset() reaches the host units as well
    chip.reset()
    orc.reset()
    _compare(chip, orc, 5)


def test_lif_soma_inside_the_message_pipeline_fails_like_the_reference(S):
    """The built-in LIF refuses a second update in one step (src/models.cpp:502-507): on such a core the first neuron
    that receives an event AND is updated by the neuron loop raises, exactly as the reference's model does."""
    arch, net = nets.host_cores(S, position="soma_inside", soma="lif")
    for grp in net._order:
        grp.set_attribute_column("leak", np.zeros(grp.count, dtype=np.int64), S.description.ATTR_INT)
    chip = S.SpikingChip(arch)
    chip.load(net)
    with pytest.raises(RuntimeError, match="multiple updates"):
        chip.run(6, "simple")


def test_plugin_synapse_and_dendrite_units(S):
    """A synapse plugin and a dendrite plugin (tests/plugins/relay_units.cpp: the arithmetic of the built-in current_based
    synapse and accumulator behind `create_<model>()`): the host replays their cores per event.  The result equals, bit
    for bit, the oracle on the twin chip with built-in units -- and the device path on that twin."""
    arch_p, net_p = nets.host_cores(S, plugin_units=True)
    arch_b, net_b = nets.host_cores(S, plugin_units=True, builtin_twin=True)
    chip = S.SpikingChip(arch_p)
    chip.load(net_p)
    orc = OracleChip(S.to_desc(arch_b, net_b))
    assert _compare(chip, orc, 40) > 200
    twin = S.SpikingChip(arch_b)  # every core on the device
    twin.load(net_b)
    chip2 = S.SpikingChip(arch_p)
    chip2.load(net_p)
    a, b = chip2.run(30, "simple"), twin.run(30, "simple")
    for ka, _ in INT_KEYS:
        assert a[ka] == b[ka], ka
    for k in DBL_KEYS:
        assert a[k] == pytest.approx(b[k], rel=1e-9), k
    assert np.array_equal(chip2.potentials(), twin.potentials())


@pytest.mark.parametrize("variant", ["many_taps", "shared_line", "inside_dendrite", "before_dendrite"])
def test_taps_dendrites_beyond_the_device_kernels(S, variant):
    """Row a21 in general (MultiTapModel1D, src/models.cpp:167-348): more than 8 taps, several synapse-receiving neurons
    on ONE unit (they share its RC line: every call of any of them advances / charges the same state), and the buffer
    inside / before the dendrite unit.  The device kernels cover <= 8 taps, one receiver per unit, `soma`/outside; such
    cores run on the host instead -- same results as the oracle, step by step."""
    kw = {"many_taps": dict(max_taps=14), "shared_line": dict(neurons_per_unit=2),
          "inside_dendrite": dict(buffer=("dendrite", True)), "before_dendrite": dict(buffer=("dendrite", False))}[variant]
    arch, net = nets.taps_dendrites(S, **kw)
    chip = S.SpikingChip(arch)
    chip.load(net)
    orc = OracleChip(S.to_desc(arch, net))
    assert _compare(chip, orc, 45) > 100


def test_host_cores_need_simple_timing_and_one_rank(S, monkeypatch):
    monkeypatch.setenv("SANAFE_HOST_CORES", "1")  # (the host replay; on the device such cores take detailed timing, below)
    arch, net = nets.host_cores(S, position="axon_out")
    chip = S.SpikingChip(arch)
    chip.load(net)
    with pytest.raises(NotImplementedError, match="simple timing"):
        chip.run(2, "detailed")
    sharded = S.SpikingChip(arch, device=0, n_ranks=2, rank=0)
    with pytest.raises(NotImplementedError, match="single-rank"):
        sharded.load(net)


@pytest.mark.parametrize("position", ["soma_inside", "axon_out"])
def test_message_pipeline_somas_in_one_batched_run(S, position):
    """The same cores through ONE sim() call with records (no host round trip between steps: the chip has no host replay
    object any more): per-step totals and the spike record rows -- which hold the status at the END of each step, as the
    reference's traces do -- against the oracle."""
    arch, net = nets.host_cores(S, position=position, seed=11)
    chip = S.SpikingChip(arch)
    chip.load(net)
    assert chip.device_layout()["msg_cores_on_device"] == 2
    orc = OracleChip(S.to_desc(arch, net))
    steps = 60
    tot = chip.run(steps, "simple", record=True)
    recs = chip.step_totals(0, steps)
    acc = {}
    for t in range(steps):
        b = orc.step("simple")
        for ka, kb in INT_KEYS:
            assert recs[ka][t] == b[kb], (t, ka)
        for k in DBL_KEYS:
            assert recs[k][t] == pytest.approx(b[k], rel=1e-9, abs=1e-30), (t, k)
        assert np.array_equal(chip.step_fired(t), (orc.status() == 3).astype(np.uint8)), t
        for k, v in b.items():
            acc[k] = acc.get(k, 0) + v
    for ka, kb in INT_KEYS:
        assert tot[ka] == acc[kb]
    assert np.array_equal(chip.potentials(), orc.potentials()) and np.array_equal(chip.status(), orc.status())


@pytest.mark.parametrize("position", ["soma_inside", "axon_out"])
def test_message_pipeline_somas_under_detailed_timing(S, position):
    """Detailed timing and the message trace of a chip whose message-pipeline somas run on the device: a message INTO such a
    core costs the axon-in latency plus, per synaptic event, the synapse's, the dendrite's and the soma's latency by the
    status that update returned (process_message, src/chip.cpp:738-789) -- msgsoma_kernel counts the updates that fired per
    message and step, the host rebuilds the messages from the neuron loop's statuses and schedules them
    (src/schedule.cpp:234-281).  Every field of every message and sim_time against the oracle; the processing delay is a sum
    the host orders by status rather than by synapse, hence the 1e-12."""
    arch, net = nets.host_cores(S, position=position, seed=5)
    chip = S.SpikingChip(arch)
    chip.load(net)
    assert chip.device_layout()["msg_cores_on_device"] == 2
    orc = OracleChip(S.to_desc(arch, net))
    into_msg_cores = fired_updates = 0
    for t in range(30):
        a = chip.run(1, "detailed", record=True, messages=True)
        b = orc.step("detailed")
        for ka, kb in INT_KEYS:
            assert a[ka] == b[kb], (t, ka, a[ka], b[kb])
        for k in DBL_KEYS:
            assert a[k] == pytest.approx(b[k], rel=1e-9, abs=1e-30), (t, k, a[k], b[k])
        ma, mb = chip.step_messages(0), orc.messages()
        assert len(ma) == len(mb), t
        for name in ma.dtype.names:
            if ma[name].dtype.kind == "f":
                fin = np.isfinite(mb[name])
                assert np.array_equal(np.isfinite(ma[name]), fin), (t, name)
                assert np.allclose(ma[name][fin], mb[name][fin], rtol=1e-12, atol=0.0), (t, name)
            else:
                assert np.array_equal(ma[name], mb[name]), (t, name)
        live = ma[ma["placeholder"] == 0]
        into = np.isin(live["dest_core_id"], [2, 4])
        into_msg_cores += int(into.sum())
        assert np.array_equal(chip.step_fired(0), (orc.status() == 3).astype(np.uint8)), t
    assert into_msg_cores > 50
    # the batched run (several steps per device call, scheduler threads) gives the same simulated time
    chip.reset()
    orc.reset()
    tot = chip.run(20, "detailed")
    ref = sum(orc.step("detailed")["sim_time"] for _ in range(20))
    assert tot["sim_time"] == pytest.approx(ref, rel=1e-9)


@pytest.mark.parametrize("position", ["soma_inside", "axon_out"])
def test_message_pipeline_somas_optional_perf_columns(S, position):
    """Optional perf columns (sim_trace_get_optional_traces, src/chip.cpp:1541-1579) of a chip whose message-pipeline somas
    run on the device: the synapse, dendrite and soma unit of such a core are charged per synaptic event, the soma by the
    status its update returned -- from the per-message fired counts the kernel logs.  Tile, core and unit columns against
    the oracle."""
    arch, net = nets.host_cores(S, position=position, seed=7)
    arch.tiles[2].log_energy = True
    cores = arch.cores()
    cores[2].log_energy = True
    cores[3].log_energy = True
    seen = set()
    for core in cores:
        for u in core.units:
            if id(u) not in seen:
                seen.add(id(u))
                u.log_energy = True
                u.log_latency = bool(u.implements & S.description.IMPL_SOMA)
    chip = S.SpikingChip(arch)
    chip.load(net)
    assert chip.device_layout()["msg_cores_on_device"] == 2
    orc = OracleChip(S.to_desc(arch, net))
    names = chip.perf_columns()
    steps = 25
    r = chip.sim(steps, timing_model="simple", perf_trace=True)
    nonzero = 0
    for t in range(steps):
        orc.step("simple")
        want = orc.optional_traces()
        assert sorted(want) == names
        for n in names:
            assert r["perf_trace"][n][t] == pytest.approx(want[n], rel=1e-12, abs=1e-30), (t, n)
            nonzero += want[n] != 0.0
    assert nonzero > 3 * steps
