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


def _run(S, arch, net, sample=None):
    chip = S.SpikingChip(arch)
    chip.load(net)
    if sample is not None:
        chip.set_state_log(sample)
    tot = chip.run(STEPS, "simple", record=True, state=sample is not None)
    recs = chip.step_totals(0, STEPS)
    fired = np.stack([chip.step_fired(t) for t in range(STEPS)])
    if sample is not None:
        return chip, tot, recs, fired, chip.potentials(), chip.step_state(0, STEPS)
    return chip, tot, recs, fired, chip.potentials()


class _Replay:
    """An independent check at full size: nothing of the product's mapper, slice cutter or device layout.
    From the RAW (src, dst, w) edge list of the recipe and the spike rows the device recorded for steps 1..K, numpy
    recomputes the synaptic input and the LIF trajectory (src/models.cpp:441-567: leak, `static_cast<int>(V * 64) / 64`,
    bias, input current, strict `>` threshold, hard reset) of ~1,000 neurons spread over ALL cores and compares their
    potentials with the device's, bit for bit, at every step.  (The dendrite is a delay line with delay 0: what arrives in
    step t reaches the soma in step t + 1.)"""

    def __init__(self, S, cores, npc, seed=1, p_fire=0.1):
        import scipy.sparse as sp
        n = cores * npc
        rng = np.random.default_rng(123)
        per_core = max(1, 1024 // cores)
        self.sample = np.sort(np.concatenate([c * npc + rng.choice(npc, per_core, replace=False) for c in range(cores)]))
        self.bias = np.where(np.random.default_rng(seed).random(n) < p_fire, 128.0, 0.0)[self.sample]  # bench.build_workload
        src, dst, w = S.chip.generate_random_edges(n, OUT_DEGREE, seed, shard=None, window=n)  # the recipe's own call
        assert len(src) == n * OUT_DEGREE
        member = np.zeros(n, dtype=bool)
        member[self.sample] = True
        rows = np.full(n, -1, dtype=np.int64)
        rows[self.sample] = np.arange(len(self.sample))
        keep = np.flatnonzero(member[dst])
        self.A = sp.csr_matrix((w[keep], (rows[dst[keep]], src[keep])), shape=(len(self.sample), n))
        self.has = sp.csr_matrix((np.ones(len(keep)), (rows[dst[keep]], src[keep])), shape=(len(self.sample), n))
        assert 0.9 * OUT_DEGREE * len(self.sample) < len(keep) < 1.1 * OUT_DEGREE * len(self.sample)  # in-degree ~ out-degree
        del src, dst, w, member, rows

    def check(self, fired, vlog, threshold=64.0, reset=0.0):
        v = np.zeros(len(self.sample))
        for t in range(fired.shape[0]):
            cur = np.zeros(len(self.sample))
            if t > 0:
                f = fired[t - 1].astype(np.float64)
                cur = self.A @ f                       # sums of integers: exact in any order
                assert np.all(cur[(self.has @ f) == 0] == 0.0)
            # loihi_leak: leak_decay 1, input_decay 0 (the current does not persist); loihi_quantize
            v = np.trunc(v * 64.0) / 64.0
            v = v + self.bias
            v = v + cur
            spike = v > threshold
            v = np.where(spike, reset, v)
            assert np.array_equal(spike, fired[t][self.sample].astype(bool)), t
            assert np.array_equal(v, vlog[t]), (t, int(np.sum(v != vlog[t])))
        return int(fired[:, self.sample].sum())


@pytest.mark.parametrize("cores,npc", [(512, 512), (1024, 256)])
def test_c3_full_size_properties(S, monkeypatch, cores, npc):
    import bench
    arch, net = bench.build_workload(S, 1, cores, npc, OUT_DEGREE, 0.1, 1)
    replay = _Replay(S, cores, npc)
    chip, tot, recs, fired, v, vlog = _run(S, arch, net, replay.sample)
    lay, info = chip.device_layout(), chip.info()
    assert info["n_synapses"] == 262144 * OUT_DEGREE
    assert replay.check(fired, vlog) > 0.15 * STEPS * len(replay.sample)  # ... of neurons that really fire
    # 16 distinct integer weights, 513 accumulators per core: 2-byte dictionary-coded words, integer accumulators;
    # the stream path is what runs
    assert lay["syn_format"] == 7 and lay["n_compact_slices"] == info["n_slices"]
    for t in range(STEPS):
        n_fired = int(fired[t].sum())
        assert recs["neurons_fired"][t] == n_fired > 0
        assert recs["neurons_updated"][t] == 262144                     # force_update: every neuron, every step
        assert recs["spikes"][t] == n_fired * OUT_DEGREE                # every neuron has exactly 2,621 out-synapses
        assert recs["packets_sent"][t] <= n_fired * cores               # at most one message per destination core
        assert recs["synapse_energy"][t] == pytest.approx(recs["spikes"][t] * 33.6e-12, rel=1e-9)  # arch/loihi_large.yaml
    assert tot["spikes"] == int(recs["spikes"].sum()) and tot["neurons_fired"] == int(fired.sum())
    # the one-step synaptic delay: nothing but the biased neurons can fire in step 1
    assert recs["neurons_fired"][0] < 0.2 * 262144 < recs["neurons_fired"][2]
    del chip

    if (cores, npc) == (1024, 256):
        # the line-of-record shape: the same steps once more with EVERY step delivered by the event kernel (the layout the
        # device falls back on when few neurons fire), against the same numpy replay and the streaming run above
        assert lay["sub_accumulators"] and lay["n_bitmap_slices"] == info["n_slices"] and lay["event_layout"] is not None
        monkeypatch.setenv("SANAFE_EVENT", "2")
        # ... through both copies of the block table: the group-major one is what the line of record's steps read (34 % of
        # the neurons fire), the neuron-major one what quiet steps read; forced modes take the latter unless told otherwise
        for table in ("group", "neuron"):
            if table == "group":
                monkeypatch.setenv("SANAFE_EVENT_SPARSE_EVENTS", "0")
            else:
                monkeypatch.delenv("SANAFE_EVENT_SPARSE_EVENTS")
            chip_e, tot_e, recs_e, fired_e, v_e, vlog_e = _run(S, arch, net, replay.sample)
            lay_e = chip_e.device_layout()
            assert lay_e["pushed_steps"] == STEPS and lay_e["event_layout"]["sparse_steps"] == (0 if table == "group" else STEPS), lay_e
            replay.check(fired_e, vlog_e)
            assert np.array_equal(fired, fired_e) and np.array_equal(v, v_e)
            for k in ("spikes", "packets_sent", "neurons_updated", "neurons_fired", "total_hops"):
                assert np.array_equal(recs[k], recs_e[k]), (table, k)
            for k in ("total_energy", "synapse_energy", "soma_energy", "network_energy", "sim_time"):
                assert np.allclose(recs[k], recs_e[k], rtol=1e-12, atol=0), (table, k)
            del chip_e
        return
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
