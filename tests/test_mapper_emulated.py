"""Host logic without a GPU: the mapper's lowered image, executed by the numpy emulation of the
device kernels (tests/image_emulator.py), must reproduce the oracle step for step."""
import numpy as np
import pytest

import nets
from image_emulator import ImageEmulator
from oracle.oracle import OracleChip

INT_KEYS = (("spikes", "spike_count"), ("packets_sent", "packets_sent"), ("neurons_updated", "neurons_updated"),
            ("neurons_fired", "neurons_fired"), ("total_hops", "total_hops"))
DBL_KEYS = ("total_energy", "synapse_energy", "dendrite_energy", "soma_energy", "network_energy", "sim_time")


def compare(S, arch, net, steps, exact_v=True):
    im, slot_of = S.map_only(arch, net)
    emu = ImageEmulator(im)
    orc = OracleChip(S.to_desc(arch, net))
    for t in range(steps):
        a, b = emu.step(), orc.step("simple")
        for ka, kb in INT_KEYS:
            assert a[ka] == b[kb], (t, ka, a[ka], b[kb])
        for k in DBL_KEYS:
            assert a[k] == pytest.approx(b[k], rel=1e-9, abs=1e-30), (t, k)
        assert np.array_equal(emu.status[slot_of], orc.status()), t
        if exact_v:
            assert np.array_equal(emu.v[slot_of], orc.potentials()), t
        else:
            assert np.allclose(emu.v[slot_of], orc.potentials(), rtol=1e-9, atol=1e-12), t


def test_example_chip(S):
    compare(S, *nets.example(S), steps=20)


def test_random_loihi_before_soma(S):
    arch, net = nets.random_loihi(S, n_tiles=2, neurons_per_core=40, out_degree=12, arch_kind="loihi", refractory=True)
    compare(S, arch, net, steps=25)


def test_random_loihi_inside_dendrite_delay(S):
    arch, net = nets.random_loihi(S, n_tiles=2, neurons_per_core=40, out_degree=12, arch_kind="large", delays=True)
    compare(S, arch, net, steps=30)


def test_random_loihi_inside_dendrite_plain_accumulator_loses_input(S):
    """with the buffer inside the dendrite unit, `accumulator` delivers 0.0 every step."""
    arch, net = nets.random_loihi(S, n_tiles=1, neurons_per_core=30, out_degree=8, arch_kind="large")
    for g in net._order:
        g.dendrite_hw[:] = net.strings("loihi_dendrites")
    compare(S, arch, net, steps=15)


def test_truenorth(S):
    compare(S, *nets.truenorth_net(S, n_tiles=6, neurons_per_core=32), steps=25)


def test_float_weights_within_tolerance(S):
    arch, net = nets.random_loihi(S, n_tiles=2, neurons_per_core=32, out_degree=10, arch_kind="loihi", weights="float")
    compare(S, arch, net, steps=15, exact_v=False)


def test_unsupported_configs_fail_loudly(S):
    arch = S.presets.example_chip(api=S.description)
    for c in arch.cores():
        c.buffer_position = S.description.BUF_BEFORE_DENDRITE
    net = S.presets.example_snn(arch)
    with pytest.raises(NotImplementedError, match="buffer position"):
        S.map_only(arch, net)


def test_unmapped_neuron_raises(S):
    arch, net = nets.example(S)
    net.groups["out"].core[1] = -1
    with pytest.raises(S.HardwareMappingError, match="not mapped"):
        S.map_only(arch, net)


def test_image_independent_of_mapper_thread_count(S, monkeypatch):
    """The edge passes of the mapper run on host threads; the lowered image must not depend on how many."""
    arch, net = nets.random_loihi(S, n_tiles=8, neurons_per_core=256, out_degree=40, arch_kind="large", delays=True, seed=11)
    images = []
    for n in ("1", "3", "8"):
        monkeypatch.setenv("SANAFE_MAP_THREADS", n)
        im, slot = S.chip.map_only(arch, net)
        images.append((im, slot))
    assert images[0][0]["n_synapses"] > 65536 * 3  # enough edges for several blocks
    for im, slot in images[1:]:
        assert np.array_equal(slot, images[0][1])
        for k, v in images[0][0].items():
            if isinstance(v, np.ndarray):
                assert np.array_equal(v, im[k]), k
            else:
                assert v == im[k], k
