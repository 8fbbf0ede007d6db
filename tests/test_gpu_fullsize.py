"""Size-independent properties at BASELINE.json's full size (configs[2]: loihi_large + 262,144 LIF neurons,
out-degree 2,621, 687 M synapses) -- the oracle cannot run this size, so the checks are properties the domain
offers: linearity of the counters in the spikes, agreement of two independent device code paths, determinism."""
import os
import sys

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

OUT_DEGREE = 2621
STEPS = int(os.environ.get("SANAFE_FULLSIZE_STEPS", "6"))  # a one-off soak run widens it (120 steps: DESIGN.md)


def _run(S, arch, net):
    chip = S.SpikingChip(arch)
    chip.load(net)
    tot = chip.run(STEPS, "simple", record=True)
    recs = chip.step_totals(0, STEPS)
    fired = np.stack([chip.step_fired(t) for t in range(STEPS)])
    return chip, tot, recs, fired, chip.potentials()


def test_c3_full_size_properties(S, monkeypatch):
    import bench
    arch, net = bench.build_workload(S, 1, 512, 512, OUT_DEGREE, 0.1, 1)
    chip, tot, recs, fired, v = _run(S, arch, net)
    lay, info = chip.device_layout(), chip.info()
    assert info["n_synapses"] == 262144 * OUT_DEGREE
    # 16 distinct integer weights, 513 accumulators per core: 2-byte dictionary-coded words, integer accumulators;
    # the stream path is what runs
    assert lay["syn_format"] == 7 and lay["n_compact_slices"] == info["n_slices"]
    for t in range(STEPS):
        n_fired = int(fired[t].sum())
        assert recs["neurons_fired"][t] == n_fired > 0
        assert recs["neurons_updated"][t] == 262144                     # force_update: every neuron, every step
        assert recs["spikes"][t] == n_fired * OUT_DEGREE                # every neuron has exactly 2,621 out-synapses
        assert recs["packets_sent"][t] <= n_fired * 512                 # at most one message per destination core
        assert recs["synapse_energy"][t] == pytest.approx(recs["spikes"][t] * 33.6e-12, rel=1e-9)  # arch/loihi_large.yaml
    assert tot["spikes"] == int(recs["spikes"].sum()) and tot["neurons_fired"] == int(fired.sum())
    # the one-step synaptic delay: nothing but the biased neurons can fire in step 1
    assert recs["neurons_fired"][0] < 0.2 * 262144 < recs["neurons_fired"][2]
    del chip

    # the same image through the other delivery code paths -- dictionary words with fp64 accumulators (format 6), 4-byte
    # int8 words streamed (format 0), 12-bit weight words gathered (format 1): identical results
    # -- and the ordered per-accumulator layout of non-integer weights (format 8), forced onto this integer network
    for force, fmt in (("6", 6), ("0", 0), ("1", 1), ("8", 8)):
        monkeypatch.setenv("SANAFE_SYN_FORMAT", force)
        chip2, tot2, recs2, fired2, v2 = _run(S, arch, net)
        assert chip2.device_layout()["syn_format"] == fmt
        assert np.array_equal(fired, fired2)
        assert np.array_equal(v, v2)                                    # integer weights: exact in any order
        for k in ("spikes", "packets_sent", "neurons_updated", "neurons_fired", "total_hops"):
            assert np.array_equal(recs[k], recs2[k]), k
        for k in ("total_energy", "synapse_energy", "soma_energy", "network_energy", "sim_time"):
            assert np.allclose(recs[k], recs2[k], rtol=1e-12, atol=0), k
        del chip2
