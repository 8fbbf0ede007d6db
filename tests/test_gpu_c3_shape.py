"""Config C3's per-core delivery shape against the oracle ("what's weak" 1).

The full-size test (test_gpu_fullsize.py) can only compare the device formats with each other; they share the mapper,
phase A of the delivery kernel, the bitmap and the slice cutter.  Here the HIP path meets the ORACLE on a network whose
destination cores look exactly like C3's: 262,144 source neurons on 512 cores, ~260 k inbound axons per destination
core with ~5 synapses each, 16 delivery slices of 16,384 axons per core (runs of 8 chunks per wavefront), write-back
shared between the slices of a core, delay-line dendrites -- in the headline's format 7 (2-byte dictionary words, integer
accumulators), format 0 (4-byte int8 words), format 4 (4-byte words + fp64 weights) and the ordered layout of
non-integer weights (format 8, forced here onto the integer network: ~2,700 entries per accumulator list)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import nets  # noqa: E402
from oracle.oracle import OracleChip  # noqa: E402

pytestmark = [pytest.mark.gpu, pytest.mark.slow]

STEPS = 10
INT_KEYS = (("spikes", "spike_count"), ("packets_sent", "packets_sent"), ("neurons_updated", "neurons_updated"),
            ("neurons_fired", "neurons_fired"), ("total_hops", "total_hops"))
DBL_KEYS = ("total_energy", "synapse_energy", "dendrite_energy", "soma_energy", "network_energy", "sim_time")


_cache = {}


def c3_shape(S, delays):
    """(arch, net, the oracle's totals / statuses / potentials of STEPS steps), built once per variant."""
    if delays not in _cache:
        _cache.clear()  # one 10.7 M-synapse network at a time
        arch, net = nets.c3_delivery_shape(S, p_fire=0.3, delays=delays)
        orc = OracleChip(S.to_desc(arch, net))
        ref = []
        for _ in range(STEPS):
            r = orc.step("simple")
            ref.append((r, orc.status(), orc.potentials()))
        del orc
        _cache[delays] = (arch, net, ref)
    return _cache[delays]


# delays=False is the headline's kernel variant (bench.py's recipe has no synaptic delays: deliver_kernel<7, false, ...>);
# delays=True adds the six accumulator rows of a delay line that is actually used (6 x 513 accumulators per core do not fit
# the 10-bit index of the dictionary words: format 0 with integer accumulators)
@pytest.mark.parametrize("delays,force,fmt", [(False, None, 7), (False, "7-delta", 7), (False, "0", 0), (False, "4", 4), (False, "8", 8),
                                              (True, None, 0), (True, "8", 8)])
def test_c3_delivery_shape_matches_the_oracle(S, monkeypatch, delays, force, fmt):
    arch, net, ref = c3_shape(S, delays)
    if force == "7-delta":  # the headline format on 2-byte delta axon records instead of the source bitmaps
        monkeypatch.setenv("SANAFE_AXON_BITMAP", "0")
        force = None
    monkeypatch.setenv("SANAFE_MIN_SLICE_AXONS", "16384")  # C3's slices: 133 M axons / 8,192 slices -> 16,384 axons each
    if force is not None:
        monkeypatch.setenv("SANAFE_SYN_FORMAT", force)
    chip = S.SpikingChip(arch)
    chip.load(net)
    lay, info = chip.device_layout(), chip.info()
    assert lay["syn_format"] == fmt
    assert info["n_neurons"] == 262144 and info["n_synapses"] == 262144 * 41
    # the C3 delivery shape: every slice on compact records, >= 16 slices on each destination core, ~5 synapses per axon
    assert lay["n_compact_slices"] == info["n_slices"] >= 8 * 16 or (delays and fmt == 8)
    assert 4.5 < info["n_synapses"] / info["n_axons"] < 5.6
    if fmt == 7:  # every destination core hears from nearly every neuron: source-bitmap axon records unless switched off
        assert lay["n_bitmap_slices"] == (0 if os.environ.get("SANAFE_AXON_BITMAP") == "0" else info["n_slices"])
    fired_any = 0
    for t in range(STEPS):
        a = chip.run(1, "simple", record=True)
        b, st, v = ref[t]
        for ka, kb in INT_KEYS:
            assert a[ka] == b[kb], (t, ka, a[ka], b[kb])
        for k in DBL_KEYS:
            assert a[k] == pytest.approx(b[k], rel=1e-9, abs=1e-30), (t, k)
        assert np.array_equal(chip.status(), st), t
        assert np.array_equal(chip.potentials(), v), t  # integer weights: exact in any order
        fired_any += int((st == 3).sum())
    # the targets must really be driven by synaptic input, not only by their biases
    dest = np.unique(np.linspace(0, 511, 8).astype(np.int64))
    dest_fired = sum(int((ref[t][1].reshape(512, 512)[dest] == 3).sum()) for t in range(STEPS))
    assert dest_fired > 0.15 * STEPS * len(dest) * 512


@pytest.mark.parametrize("sub", [True, False])
def test_survey_shape_1024_cores_of_256_matches_the_oracle(S, monkeypatch, sub):
    """SURVEY 8(d)'s shape of C3 -- the bench default: 1,024 cores x 256 neurons -- with 16 destination cores that hear from
    (nearly) every one of the 262,144 neurons: bitmap axon records, cores cut into slices at multiples of 8,192 source slots,
    and, on cores of at most 256 neurons, the SUB-ACCUMULATOR instantiation of the delivery kernel (16 partial accumulators per
    neuron, picked by the weight code; `SANAFE_SUB_ACCUMULATORS=0` keeps one per neuron).  Ten steps against the oracle."""
    _cache.clear()
    arch, net = nets.c3_delivery_shape(S, cores=1024, neurons_per_core=256, dest_cores=16, out_degree=41, p_fire=0.3, delays=False)
    orc = OracleChip(S.to_desc(arch, net))
    monkeypatch.setenv("SANAFE_MIN_SLICE_AXONS", "16384")
    if not sub:
        monkeypatch.setenv("SANAFE_SUB_ACCUMULATORS", "0")
    chip = S.SpikingChip(arch)
    chip.load(net)
    lay, info = chip.device_layout(), chip.info()
    assert lay["syn_format"] == 7 and lay["n_bitmap_slices"] == info["n_slices"] >= 16 * 4
    assert lay["sub_accumulators"] == sub
    fired = 0
    for t in range(STEPS):
        a, b = chip.run(1, "simple", record=True), orc.step("simple")
        for ka, kb in INT_KEYS:
            assert a[ka] == b[kb], (t, ka, a[ka], b[kb])
        for k in DBL_KEYS:
            assert a[k] == pytest.approx(b[k], rel=1e-9, abs=1e-30), (t, k)
        st = orc.status()
        assert np.array_equal(chip.status(), st), t
        assert np.array_equal(chip.potentials(), orc.potentials()), t
        fired += int((st == 3).sum())
    assert fired > 0.3 * STEPS * 262144 * 0.9
