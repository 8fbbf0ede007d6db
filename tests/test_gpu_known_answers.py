"""The reference's own model-level known answers (tests/unit/*.cpp), asserted on the HIP path directly: each case is
restated as a one-core network run through ``SpikingChip.sim()`` and checked against the LITERAL value the reference
test expects -- not against the oracle.

A unit test hands a current to ``update(addr, current, t)``; through the chip the same current reaches the soma as
the weight of a synapse from an input neuron that spikes at step 1, i.e. one timestep later, so
"update at time t" of a unit test is timestep t + 1 here.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

IDLE, UPDATED, FIRED = 1, 2, 3  # NeuronStatus (src/mapped.hpp:22-28)

SOMA_COSTS = {"energy_access_neuron": 0.0, "latency_access_neuron": 0.0, "energy_update_neuron": 0.0,
              "latency_update_neuron": 0.0, "energy_spike_out": 0.0, "latency_spike_out": 0.0}


def one_core(S, soma_model="leaky_integrate_fire", dendrite="accumulator", n_inputs=4):
    D = S.description
    arch = D.Architecture("unit", 1, 1, 4)
    tile = arch.create_tile("tile[0]")
    core = arch.create_core("core[0]", tile.id, "soma", False, 64)
    core.create_axon_in("axon_in", 0.0, 0.0)
    core.create_synapse("synapse", "current_based", {"energy_process_spike": 0.0, "latency_process_spike": 0.0})
    core.create_dendrite("dendrite", dendrite, {"energy_update": 0.0, "latency_update": 0.0})
    core.create_soma("soma", soma_model, dict(SOMA_COSTS))
    for i in range(n_inputs):
        core.create_soma("input[%d]" % i, "input", dict(SOMA_COSTS))
    core.create_axon_out("axon_out", 0.0, 0.0)
    return arch, core


def drive(S, arch, core, neuron_attrs, currents, steps, soma_model="leaky_integrate_fire", tap=None):
    """One neuron under test + one input neuron per entry of `currents` = (unit-test time t, weight): the input
    spikes at timestep t so the weight arrives at t + 1.  Returns per-step (status, potential) of the neuron."""
    D = S.description
    net = D.Network("ka")
    n_in = max(1, len(currents))
    gin = net.create_neuron_group("in", n_in, {}, "synapse", "dendrite", False, False)
    for i in range(n_in):
        train = [0.0] * steps
        if i < len(currents):
            train[currents[i][0] - 1] = 1.0
        gin.apply_config(i, i + 1, soma_hw_name="input[%d]" % i, attrs={"spikes": ((D.ATTR_LIST, 0.0, None, train), D.FWD_ALL)})
    g = net.create_neuron_group("n", 1, neuron_attrs, "synapse", "dendrite", True, True, "soma")
    if currents:
        pairs = np.array([[i, 0] for i in range(len(currents))])
        attrs = {"weight": [float(w) for _, w in currents]}
        if tap is not None:
            attrs["tap"] = [int(tap)] * len(currents)
        gin.connect_neurons_sparse(g, attrs, pairs, narrow_float=False)
    gin.map_to_core(core, 0, n_in)
    g.map_to_core(core, 0, 1)
    chip = S.SpikingChip(arch)
    chip.load(net)
    gid = chip._built.groups["n"][0]
    out = []
    for _ in range(steps):
        chip.run(1, "simple")
        out.append((int(chip.status()[gid]), float(chip.potentials()[gid])))
    return out


LIF = {"threshold": 64.0, "reset": 0.0, "reset_mode": "hard", "leak_decay": 1.0, "input_decay": 0.0, "bias": 0.0,
       "force_update": False}


def test_lif_fires_when_above_threshold(S):
    # tests/unit/test_loihi_lif.cpp:28-45: update(0, 80.0, 1) -> fired, potential 0.0
    arch, core = one_core(S)
    trace = drive(S, arch, core, LIF, [(1, 80.0)], 3)
    assert trace[1][0] == FIRED and trace[1][1] == pytest.approx(0.0, abs=1e-6)


def test_lif_does_not_fire_below_threshold_and_stays_stable(S):
    # :47-64 update(0, 50.0, 1) -> updated, potential 50; :66-85 then update(0, nullopt, 2) -> updated, potential 50
    arch, core = one_core(S)
    trace = drive(S, arch, core, LIF, [(1, 50.0)], 4)
    assert trace[0] == (IDLE, 0.0)  # nothing yet: V = 0, no input, no bias, no force_update
    assert trace[1][0] == UPDATED and trace[1][1] == pytest.approx(50.0, abs=1e-6)
    assert trace[2][0] == UPDATED and trace[2][1] == pytest.approx(50.0, abs=1e-6)


def test_lif_force_update(S):
    # :318-327 force_update, update(0, nullopt, 1) -> updated
    arch, core = one_core(S)
    trace = drive(S, arch, core, {"force_update": True}, [], 2)
    assert trace[0][0] == UPDATED and trace[1][0] == UPDATED


def test_accumulator_integrates_and_accumulates(S):
    # tests/unit/test_accumulator.cpp:22-39: 5.0 alone reads back 5.0; 2.0 + 3.0 in one step read back 5.0
    arch, core = one_core(S)
    one = drive(S, arch, core, LIF, [(1, 5.0)], 3)
    two = drive(S, arch, core, LIF, [(1, 2.0), (1, 3.0)], 3)
    assert one[1][1] == 5.0 and two[1][1] == 5.0  # EXPECT_DOUBLE_EQ


def test_current_based_synapse_weight(S):
    # tests/unit/test_current_based_synapse.cpp:22-28: weight 1.23 read back within 1e-6 (fp64 weight path)
    arch, core = one_core(S)
    trace = drive(S, arch, core, LIF, [(1, 1.23)], 3)
    assert trace[1][1] == pytest.approx(1.23, abs=1e-6)


def test_truenorth_fires_and_leaks(S):
    # tests/unit/test_truenorth.cpp:44-52: threshold 0.5, hard reset to 0, current 1.0 -> fired
    arch, core = one_core(S, "truenorth")
    trace = drive(S, arch, core, {"threshold": 0.5, "reset_mode": "hard", "reset": 0.0}, [(1, 1.0)], 3)
    assert trace[1][0] == FIRED and trace[1][1] == 0.0
    # :54-62: threshold 10, leak 0.5 towards zero, current 2.0 -> potential >= 0 (2.0: no leak is applied at V = 0)
    arch, core = one_core(S, "truenorth")
    trace = drive(S, arch, core, {"threshold": 10.0, "leak": 0.5, "leak_towards_zero": True}, [(1, 2.0)], 4)
    assert trace[1][1] == 2.0 and trace[2][1] == 1.5


def test_truenorth_random_mask(S):
    # :175-186: srand(1), threshold 1.0, random_mask 0xFF, no input -> fired at the first update
    # (the first rand() of a fresh process is 1804289383; & 0xff = 103 >= 1)
    arch, core = one_core(S, "truenorth")
    trace = drive(S, arch, core, {"threshold": 1.0, "reset_mode": "hard", "reset": 0.0, "random_mask": 0xFF}, [], 1)
    assert trace[0][0] == FIRED


def test_input_model(S):
    # tests/unit/test_inputmodel.cpp:36-97 through a chip of input neurons
    D = S.description
    arch, core = one_core(S, n_inputs=4)
    net = D.Network("inputs")
    g = net.create_neuron_group("in", 4, {}, "synapse", "dendrite", False, True)
    cases = [{"spikes": ((D.ATTR_LIST, 0.0, None, [1.0]), D.FWD_ALL)},    # :36-42 spikes {true} -> fired
             {"spikes": ((D.ATTR_LIST, 0.0, None, [0.0]), D.FWD_ALL)},    # :44-50 spikes {false} -> idle
             {"poisson": ((D.ATTR_DOUBLE, 1.0, None, None), D.FWD_ALL)},   # :79-86 poisson 1.0 -> fired (U in [0, 1))
             {"rate": ((D.ATTR_DOUBLE, 1.0, None, None), D.FWD_ALL)}]      # :88-95 rate 1.0 -> fired
    for i, attrs in enumerate(cases):
        g.apply_config(i, i + 1, soma_hw_name="input[%d]" % i, attrs=attrs)
    g.map_to_core(core, 0, 4)
    chip = S.SpikingChip(arch)
    chip.load(net)
    chip.run(1, "simple")
    assert list(chip.status()) == [FIRED, IDLE, FIRED, FIRED]
    chip.run(1, "simple")
    assert list(chip.status())[:2] == [IDLE, IDLE]  # the trains are consumed


def test_multitap_input_current_adds(S):
    # tests/unit/test_multitap.cpp:99-110: 2 taps, time constants {1, 1}, space constant {0}: an input of 1.5 at
    # tap 0 reads back 1.5
    D = S.description
    arch, core = one_core(S, dendrite="taps")
    attrs = dict(LIF)
    attrs.update({"taps": 2})
    net_attrs = {k: v for k, v in attrs.items()}
    trace = drive_taps(S, arch, core, net_attrs, [(1, 1.5)], 3)
    assert trace[1][1] == 1.5


def drive_taps(S, arch, core, neuron_attrs, currents, steps):
    D = S.description
    net = D.Network("taps")
    gin = net.create_neuron_group("in", 1, {}, "synapse", "dendrite", False, False)
    train = [0.0] * steps
    train[currents[0][0] - 1] = 1.0
    gin.apply_config(0, 1, soma_hw_name="input[0]", attrs={"spikes": ((D.ATTR_LIST, 0.0, None, train), D.FWD_ALL)})
    g = net.create_neuron_group("n", 1, {k: v for k, v in neuron_attrs.items() if k != "taps"}, "synapse", "dendrite", True, True, "soma")
    g.apply_config(0, 1, attrs={"taps": ((D.ATTR_INT, 2.0, None, None), D.FWD_ALL),
                                "time_constants": ((D.ATTR_LIST, 0.0, None, [1.0, 1.0]), D.FWD_ALL),
                                "space_constants": ((D.ATTR_LIST, 0.0, None, [0.0]), D.FWD_ALL)})
    gin.connect_neurons_sparse(g, {"weight": [currents[0][1]], "tap": [0]}, np.array([[0, 0]]), narrow_float=False)
    gin.map_to_core(core, 0, 1)
    g.map_to_core(core, 0, 1)
    chip = S.SpikingChip(arch)
    chip.load(net)
    gid = chip._built.groups["n"][0]
    out = []
    for _ in range(steps):
        chip.run(1, "simple")
        out.append((int(chip.status()[gid]), float(chip.potentials()[gid])))
    return out
