"""The C++17 / PyBind11 front-end (sanafecpp_amd: YAML subset reader, description objects, lowering) must
produce the same flat description as the pure-Python twin (description.py / yaml_io.py on PyYAML)."""
import ctypes as C
import glob
import json
import os

import numpy as np
import pytest
import yaml

import nets
from conftest import REFERENCE, ROOT, have_reference

GOLDEN = os.path.join(ROOT, "tests", "golden")


class View:
    """ctypes view of a lowered description, whichever front-end built it."""

    def __init__(self, S, built):
        self.keep = built
        self.desc = S.description.Desc.from_address(built.address) if hasattr(built, "address") else built.desc


def same_desc(S, a, b, sample=37):
    va, vb = View(S, a), View(S, b)
    da, db = va.desc, vb.desc
    for f in ("noc_width", "noc_height", "noc_buffer_size", "n_sync", "n_tiles", "n_cores", "n_templates", "n_units", "n_groups",
              "n_neurons", "n_edges"):
        assert getattr(da, f) == getattr(db, f), f
    n, e = da.n_neurons, da.n_edges

    def arr(d, f, cnt):
        return np.ctypeslib.as_array(getattr(d, f), shape=(cnt,)) if cnt else np.zeros(0)

    for f, cnt in (("neuron_core", n), ("neuron_map_order", n), ("neuron_log_spikes", n), ("neuron_log_potential", n),
                   ("edge_src", e), ("edge_dst", e), ("edge_weight", e)):
        assert np.array_equal(arr(da, f, cnt), arr(db, f, cnt)), f
    A, B = S.description.describe(va), S.description.describe(vb)
    assert A["noc"] == B["noc"] and A["sync"] == B["sync"] and A["tiles"] == B["tiles"]
    assert A["cores"][:8] == B["cores"][:8] and A["cores"][-1] == B["cores"][-1]
    assert [g["name"] for g in A["groups"]] == [g["name"] for g in B["groups"]]
    for ga, gb in zip(A["groups"], B["groups"]):
        assert ga["neurons"][::sample] == gb["neurons"][::sample]
        assert ga["neurons"][-1] == gb["neurons"][-1]
    step = max(1, len(A["edges"]) // 5000)
    assert A["edges"][::step] == B["edges"][::step]


def norm(x):
    if isinstance(x, dict):
        return {str(k): norm(v) for k, v in x.items()}
    if isinstance(x, list):
        return [norm(v) for v in x]
    return None if x is None else str(x)


def yaml_json(S, path):
    L = S.chip.lib()
    L.sanafe_yaml_file_to_json.restype = C.c_void_p
    L.sanafe_yaml_file_to_json.argtypes = [C.c_char_p]
    L.sanafe_free.argtypes = [C.c_void_p]
    p = L.sanafe_yaml_file_to_json(path.encode())
    assert p, L.sanafe_last_error()
    out = json.loads(C.string_at(p).decode())
    L.sanafe_free(p)
    return out


def yaml_files():
    files = [os.path.join(GOLDEN, "mini_arch.yaml"), os.path.join(GOLDEN, "mini_snn.yaml")]
    if have_reference():
        for pat in ("arch/*.yaml", "snn/*.yaml", "sanafe/examples/*.yaml"):
            files += sorted(glob.glob(os.path.join(REFERENCE, pat)))
    return files


@pytest.mark.parametrize("path", yaml_files())
def test_yaml_subset_reader_matches_pyyaml(S, path):
    assert yaml_json(S, path) == norm(yaml.load(open(path), Loader=S.yaml_io._Loader))


def test_yaml_subset_reader_errors(S, tmp_path):
    bad = tmp_path / "bad.yaml"
    bad.write_text("a:\n  - [1, 2\n")
    L = S.chip.lib()
    L.sanafe_yaml_file_to_json.restype = C.c_void_p
    L.sanafe_yaml_file_to_json.argtypes = [C.c_char_p]
    assert not L.sanafe_yaml_file_to_json(str(bad).encode())
    assert b"unterminated flow collection" in L.sanafe_last_error()


def test_mini_yaml_cpp_equals_twin(S):
    a_arch = S.load_arch(os.path.join(GOLDEN, "mini_arch.yaml"))
    a_net = S.load_net(os.path.join(GOLDEN, "mini_snn.yaml"), a_arch)
    b_arch = S.yaml_io.load_arch(os.path.join(GOLDEN, "mini_arch.yaml"))
    b_net = S.yaml_io.load_net(os.path.join(GOLDEN, "mini_snn.yaml"), b_arch)
    same_desc(S, S.cpp.to_desc(a_arch, a_net), S.to_desc(b_arch, b_net), sample=1)


@pytest.mark.skipif(not have_reference(), reason="reference not present")
@pytest.mark.parametrize("arch_file,snn_file", [
    ("arch/example_chip.yaml", "snn/example_snn.yaml"),
    ("arch/loihi.yaml", "snn/dvs.yaml"),
    ("arch/loihi.yaml", "snn/conv.yaml"),
    ("arch/loihi.yaml", "snn/input_net.yaml"),
])
def test_reference_yaml_cpp_equals_twin(S, arch_file, snn_file):
    try:
        b_arch = S.yaml_io.load_arch(os.path.join(REFERENCE, arch_file))
        b_net = S.yaml_io.load_net(os.path.join(REFERENCE, snn_file), b_arch)
    except Exception as exc:  # the pair does not load in the twin either: both must refuse it
        with pytest.raises(Exception):
            a = S.load_arch(os.path.join(REFERENCE, arch_file))
            S.load_net(os.path.join(REFERENCE, snn_file), a)
        pytest.skip("pair not loadable: %s" % exc)
    a_arch = S.load_arch(os.path.join(REFERENCE, arch_file))
    a_net = S.load_net(os.path.join(REFERENCE, snn_file), a_arch)
    same_desc(S, S.cpp.to_desc(a_arch, a_net), S.to_desc(b_arch, b_net))


def build_tutorial5_cpp(S):
    """tests/nets.py::tutorial5_dvs again, through the C++ API (the calls sanafe/layers.py makes)."""
    arch = S.presets.loihi()  # C++ objects by default
    d = np.load(os.path.join(GOLDEN, "dvs_challenge.npz"))
    th = d["thresholds"]
    net = S.Network()
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
    for n, b in zip(g0, d["inputs"]):
        n.set_attributes(model_attributes={"bias": b})  # exactly the notebook's loop
    cores = arch.cores()
    for n in g0:
        n.map_to_core(arch.tile_cores(0)[0])
    k = 0
    for (grp, _, _, _), cc in zip(layers, (1, 4, 16, 16, 4, 1)):
        per = len(grp) // cc
        for idx in range(cc):
            lo, hi = idx * per, (len(grp) if idx == cc - 1 else (idx + 1) * per)
            for neuron in grp[lo:hi]:
                neuron.map_to_core(cores[k])
            k += 1
    return arch, net


@pytest.mark.slow
def test_api_built_network_cpp_equals_twin(S):
    a_arch, a_net = build_tutorial5_cpp(S)
    b_arch, b_net = nets.tutorial5_dvs(S)
    assert (a_net.neuron_count, a_net.edge_count) == (18678, 3564441)
    same_desc(S, S.cpp.to_desc(a_arch, a_net), S.to_desc(b_arch, b_net))


def test_cpp_api_errors(S):
    net = S.Network()
    g = net.create_neuron_group("a", 4)
    with pytest.raises(ValueError, match="already exists"):
        net.create_neuron_group("a", 2)
    with pytest.raises(ValueError, match="Reserved neuron attribute"):
        g[0].set_attributes(model_attributes={"log_spikes": True})
    with pytest.raises(ValueError, match="dest nid is out of range"):
        g.connect_neurons_sparse(g, {"w": [1.0]}, [(0, 9)])
    with pytest.raises(IndexError):
        g[4]


def test_oracle_runs_on_cpp_description(S):
    """The oracle consumes the C++ front-end's description as well (same sanafe_desc format)."""
    from oracle.oracle import OracleChip
    a_arch = S.load_arch(os.path.join(GOLDEN, "mini_arch.yaml"))
    a_net = S.load_net(os.path.join(GOLDEN, "mini_snn.yaml"), a_arch)
    b_arch = S.yaml_io.load_arch(os.path.join(GOLDEN, "mini_arch.yaml"))
    b_net = S.yaml_io.load_net(os.path.join(GOLDEN, "mini_snn.yaml"), b_arch)
    x, y = OracleChip(S.cpp.to_desc(a_arch, a_net)), OracleChip(S.to_desc(b_arch, b_net))
    fired = 0
    for t in range(40):
        rx, ry = x.step("detailed"), y.step("detailed")
        assert rx == ry
        fired += rx["neurons_fired"]
    assert fired > 10
