"""Pin the oracle's restated unit models bit-for-bit against the
REFERENCE's own model translation units, compiled unmodified into oracle/_ref
(recipe: oracle/Makefile).  Skipped when oracle/_ref has not been built."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ref_models_lib
from unitmodels import ROOT, Unit, oracle_lib, reference_lib

pytestmark = pytest.mark.skipif(not os.path.exists(ref_models_lib()), reason="oracle/_ref not built")


@pytest.fixture(scope="module")
def libs():
    return reference_lib(), oracle_lib()


def pair(libs, model, plugin=None):
    ref, orc = libs
    return Unit(ref, model, plugin), Unit(orc, model)


def same(a, b):
    assert a.key() == b.key(), (a.key(), b.key())


def bits(x):
    return np.float64(x).tobytes()


def test_current_based_synapse(libs):
    r, o = pair(libs, "current_based")
    rng = np.random.default_rng(1)
    w = rng.normal(size=300)
    for i in rng.permutation(300):
        for u in (r, o):
            u.set_edge(int(i), "weight" if i % 2 else "w", float(w[i]))
    for i in range(300):
        same(r.syn(i, True, 1), o.syn(i, True, 1))
        same(r.syn(i, False, 1), o.syn(i, False, 1))


def test_accumulator(libs):
    r, o = pair(libs, "accumulator")
    rng = np.random.default_rng(2)
    t = 1
    for _ in range(5000):
        n = int(rng.integers(0, 16))
        cur = None if rng.random() < 0.2 else float(rng.normal())
        if rng.random() < 0.1:
            t += int(rng.integers(1, 3))
        same(r.dend(n, cur, 0, t), o.dend(n, cur, 0, t))
    with pytest.raises(RuntimeError):
        o.dend(1024, 1.0, 0, t)  # .at() past the 1024 slots (src/models.hpp:61-63)
    with pytest.raises(RuntimeError):
        r.dend(1024, 1.0, 0, t)


def test_accumulator_reset(libs):
    r, o = pair(libs, "accumulator")
    for u in (r, o):
        u.dend(3, 2.0, 0, 1)
        u.reset()
    same(r.dend(3, None, None, 1), o.dend(3, None, None, 1))
    same(r.dend(3, 1.0, None, 1), o.dend(3, 1.0, None, 1))


def test_accumulator_with_delay(libs):
    r, o = pair(libs, "accumulator_with_delay")
    rng = np.random.default_rng(3)
    for s in range(40):
        d = int(rng.integers(0, 6))
        for u in (r, o):
            u.set_edge(s, "delay" if s % 2 else "d", d)
    for u in (r, o):
        with pytest.raises(RuntimeError):
            u.set_edge(41, "delay", 6)
    t = 1
    for _ in range(8000):
        n = int(rng.integers(0, 8))
        cur = None if rng.random() < 0.3 else float(rng.normal())
        syn = None if rng.random() < 0.1 else int(rng.integers(0, 60))
        if rng.random() < 0.15:
            t += int(rng.integers(1, 4))
        same(r.dend(n, cur, syn, t), o.dend(n, cur, syn, t))


@pytest.mark.parametrize("seed", range(6))
def test_loihi_lif_random(libs, seed):
    r, o = pair(libs, "leaky_integrate_fire")
    rng = np.random.default_rng(100 + seed)
    n = 12
    modes = ["none", "soft", "hard", "saturate"]
    for i in range(n):
        attrs = {
            "threshold": float(rng.uniform(0.5, 20)), "reverse_threshold": float(-rng.uniform(0.5, 20)),
            "reset": float(rng.uniform(-1, 1)), "reverse_reset": float(rng.uniform(-3, 0)),
            "reset_mode": modes[int(rng.integers(0, 4))], "reverse_reset_mode": modes[int(rng.integers(0, 4))],
            "leak_decay": float(rng.uniform(0.8, 1.0)), "input_decay": float(rng.choice([0.0, 0.5, 0.9])),
            "bias": float(rng.choice([0.0, 0.0, 0.3, -0.2, 5.0])), "refractory_delay": int(rng.integers(0, 4)),
            "force_update": bool(rng.random() < 0.3), "log_u": bool(rng.random() < 0.5),
        }
        if rng.random() < 0.3:
            attrs["potential"] = float(rng.normal())
        for k, v in attrs.items():
            for u in (r, o):
                u.set_neuron(i, k, v)
    for t in range(1, 400):
        for i in range(n):
            cur = None if rng.random() < 0.4 else float(rng.normal() * 4)
            same(r.soma(i, cur, t), o.soma(i, cur, t))
            assert bits(r.potential(i)) == bits(o.potential(i))
            assert r.trace(i, "u") == o.trace(i, "u")
        if t == 200:
            r.reset()
            o.reset()


def test_loihi_lif_errors(libs):
    for u in pair(libs, "leaky_integrate_fire"):
        u.soma(0, None, 1)
        with pytest.raises(RuntimeError, match="multiple updates"):
            u.soma(0, None, 1)  # tests/unit/test_loihi_lif.cpp:190-200
        with pytest.raises(RuntimeError, match="every time-step"):
            u.soma(0, None, 5)  # tests/unit/test_loihi_lif.cpp:202-211


def test_loihi_noise_stream(libs, tmp_path):
    noise = tmp_path / "noise.csv"
    rng = np.random.default_rng(7)
    noise.write_text("\n".join(str(int(v)) for v in rng.integers(0, 512, size=37)) + "\n")
    r, o = pair(libs, "leaky_integrate_fire")
    for u in (r, o):
        u.set_hw("noise", str(noise))
        u.set_hw("noise_bits", 6)
        u.set_neuron(0, "threshold", 50.0)
        u.set_neuron(1, "threshold", 30.0)
    for t in range(1, 60):
        for i in range(2):
            same(r.soma(i, 0.5, t), o.soma(i, 0.5, t))
            assert bits(r.potential(i)) == bits(o.potential(i))


@pytest.mark.parametrize("seed", range(4))
def test_truenorth_random(libs, seed):
    rng = np.random.default_rng(200 + seed)
    n = 10
    modes = ["none", "soft", "hard", "saturate"]
    cfg = []
    for i in range(n):
        cfg.append({
            "threshold": float(rng.integers(1, 12)), "reverse_threshold": float(-rng.integers(1, 12)),
            "reset": float(rng.integers(-2, 2)), "reverse_reset": float(rng.integers(-3, 1)),
            "reset_mode": modes[int(rng.integers(0, 4))], "reverse_reset_mode": modes[int(rng.integers(0, 4))],
            "leak": float(rng.choice([0.0, 0.5, 1.0])), "leak_towards_zero": bool(rng.random() < 0.5),
            "bias": float(rng.choice([0.0, 1.0, -1.0])), "force_update": bool(rng.random() < 0.3),
            "random_mask": int(rng.choice([0, 0, 3, 7])),
        })
    inputs = [[None if rng.random() < 0.4 else float(rng.integers(-4, 5)) for _ in range(n)] for _ in range(300)]
    libc = ctypes.CDLL(None)
    outs = []
    for u in pair(libs, "truenorth"):
        libc.srand(1)  # the model draws from the process-wide libc stream (src/models.cpp:757)
        for i, c in enumerate(cfg):
            for k, v in c.items():
                u.set_neuron(i, k, v)
        rec = []
        for t, row in enumerate(inputs, 1):
            for i, cur in enumerate(row):
                rec.append((u.soma(i, cur, t).key(), bits(u.potential(i))))
        outs.append(rec)
    assert outs[0] == outs[1]


def test_input_model(libs):
    # seeds follow instance creation order (src/models.hpp:347, 366): create the same number on both sides
    rs = [Unit(libs[0], "input") for _ in range(3)]
    os_ = [Unit(libs[1], "input") for _ in range(3)]
    for u in (rs[0], os_[0]):
        u.set_neuron(0, "spikes", [1, 0, 1, 1, 0])
    for u in (rs[1], os_[1]):
        u.set_neuron(0, "poisson", 0.3)
    for u in (rs[2], os_[2]):
        u.set_neuron(0, "rate", 0.25)
        u.set_neuron(0, "poisson", 0.05)
    for t in range(1, 200):
        for r, o in zip(rs, os_):
            same(r.soma(0, None, t), o.soma(0, None, t))
    for u in (rs[0], os_[0]):
        with pytest.raises(RuntimeError, match="Current sent to input neuron"):
            u.soma(0, 1.0, 300)
        u.soma(0, 0.0, 300)  # zero current is accepted


def test_hodgkin_huxley_plugin(libs):
    plugin = os.path.join(ROOT, "oracle", "_ref", "libhodgkin_huxley_ref.so")
    if not os.path.exists(plugin):
        pytest.skip("HH plugin not built")
    for cur in (0.0, 100.0, 200.0):
        r = Unit(libs[0], "hodgkin_huxley", plugin)
        o = Unit(libs[1], "hodgkin_huxley")
        for u in (r, o):
            for k, v in (("m", 0.0529), ("n", 0.3177), ("h", 0.5961), ("current", cur)):
                u.set_neuron(0, k, v)
        for t in range(1, 400):
            a, b = r.soma(0, None, t), o.soma(0, None, t)
            assert a.status == b.status
            # same libm, same expression order: bit-exact on this host
            assert bits(r.potential(0)) == bits(o.potential(0))
