"""Known answers of the reference's own parser tests (tests/unit/test_yaml_arch.cpp, test_yaml_snn.cpp) asserted on
the C++17 YAML reader of the product (sana-fe_amd/host/yaml_subset.cpp + description.cpp), through load_arch /
load_net and the lowered description.  The YAML snippets are the fixtures those tests embed (inputs), the asserted
values are the literals they expect."""
import os

import pytest

from test_cpp_frontend import View

CORE_BODY = """
          attributes:
            buffer_position: soma
            max_neurons_supported: 10
          axon_in:
            - name: axin
              attributes:
                energy_message_in: 0.0
                latency_message_in: 0.0
          synapse:
            - name: syn
              attributes:
                model: current_based
                energy_process_spike: 1.0
                latency_process_spike: 1.0
          dendrite:
            - name: dend
              attributes:
                model: accumulator
                energy_update: 0.0
                latency_update: 0.0
                update_every_timestep: true
          soma:
            - name: soma
              attributes:
                model: leaky_integrate_fire
                energy_access_neuron: 1.0
                latency_access_neuron: 1.0
                energy_update_neuron: 1.0
                latency_update_neuron: 1.0
                energy_spike_out: 1.0
                latency_spike_out: 1.0
          axon_out:
            - name: axout
              attributes:
                energy_message_out: 1.0
                latency_message_out: 1.0
"""
TILE_ATTRS = """
      attributes:
        energy_north_hop: 1.0
        latency_north_hop: 1.0
        energy_east_hop: 1.0
        latency_east_hop: 1.0
        energy_south_hop: 1.0
        latency_south_hop: 1.0
        energy_west_hop: 1.0
        latency_west_hop: 1.0
"""


def arch_yaml(name, width, tile, core):
    return ("architecture:\n  name: %s\n  attributes:\n    link_buffer_size: 1\n    width: %d\n    height: 1\n  tile:\n"
            "    - name: %s%s      core:\n        - name: %s%s" % (name, width, tile, TILE_ATTRS, core, CORE_BODY))


def load(S, tmp_path, text, fname="arch.yaml"):
    p = tmp_path / fname
    p.write_text(text)
    return S.load_arch(str(p))


def described(S, arch, net=None):
    net = net if net is not None else S.Network("empty")
    return S.description.describe(View(S, S.cpp.to_desc(arch, net)))


def test_parses_basic_architecture(S, tmp_path):
    # tests/unit/test_yaml_arch.cpp:149-288
    arch = load(S, tmp_path, arch_yaml("minimal_arch", 1, "tile0", "core0"))
    assert len(arch.tiles) == 1 and arch.core_count == 1 and arch.name == "minimal_arch"
    assert (arch.noc_width, arch.noc_height, arch.noc_buffer_size) == (1, 1, 1)
    assert arch.cores()[0].name == "core0[0]"  # names always get an index suffix (src/yaml_arch.cpp:392-395)
    core = described(S, arch)["cores"][0]
    assert core["axon_in"] == [(0.0, 0.0)] and core["axon_out"] == [(1.0, 1.0)]
    units = core["units"]
    assert [u["name"] for u in units] == ["syn", "dend", "soma"]  # parsed synapse -> dendrite -> soma
    assert [u["model"] for u in units] == ["current_based", "accumulator", "leaky_integrate_fire"]
    assert [u["impl"] for u in units] == [S.description.IMPL_SYNAPSE, S.description.IMPL_DENDRITE, S.description.IMPL_SOMA]
    assert units[0]["attrs"]["energy_process_spike"][1] == 1.0 and units[0]["attrs"]["latency_process_spike"][1] == 1.0
    assert units[1]["attrs"]["energy_update"][1] == 0.0 and units[1]["attrs"]["latency_update"][1] == 0.0
    assert units[1]["flags"] & S.description.UNIT_UPDATE_EVERY_TIMESTEP
    for k in ("energy_access_neuron", "latency_access_neuron", "energy_update_neuron", "latency_update_neuron",
              "energy_spike_out", "latency_spike_out"):
        assert units[2]["attrs"][k][1] == 1.0


def test_tile_and_core_range_notation(S, tmp_path):
    # :290-368 tile[0..2] expands to three tiles; :370-447 core[0..3] to four cores
    arch = load(S, tmp_path, arch_yaml("range_test_arch", 3, "tile[0..2]", "core0"))
    assert [t.name for t in arch.tiles] == ["tile[0]", "tile[1]", "tile[2]"] and arch.core_count == 3
    arch = load(S, tmp_path, arch_yaml("core_range", 1, "tile0", "core[0..3]"))
    assert len(arch.tiles) == 1 and [c.name for c in arch.cores()] == ["core[0]", "core[1]", "core[2]", "core[3]"]


def test_missing_sections_throw(S, tmp_path):
    # :449-559 an architecture without tile / core / soma sections is rejected; :561-566 so is a missing file
    good = arch_yaml("a", 1, "tile0", "core0")
    no_tile = good[:good.index("  tile:")]
    no_core = good[:good.index("      core:")]
    no_soma = good.replace(good[good.index("          soma:"):good.index("          axon_out:")], "")
    for text in (no_tile, no_core, no_soma):
        with pytest.raises(Exception):
            load(S, tmp_path, text)
    with pytest.raises(Exception):
        S.load_arch(str(tmp_path / "nonexistent.yaml"))


NET_HEAD = "network:\n"


def load_net(S, tmp_path, arch, body, mappings="mappings: []\n"):
    p = tmp_path / "net.yaml"
    p.write_text("network:\n" + body + mappings)
    return S.load_net(str(p), arch)


def test_parse_full_network_section(S, tmp_path):
    # tests/unit/test_yaml_snn.cpp:187-229
    arch = load(S, tmp_path, arch_yaml("a", 1, "tile0", "core0"))
    net = load_net(S, tmp_path, arch, """  name: example
  groups:
    - name: Input
      neurons:
        - 0..1
    - name: Output
      neurons:
        - 0..1
  edges:
    - Input.0 -> Output.0: [weight: -1.0]
    - Input.1 -> Output.1: [weight: -2.0]
""")
    d = described(S, arch, net)
    assert sorted(g["name"] for g in d["groups"]) == ["Input", "Output"]
    assert all(len(g["neurons"]) == 2 for g in d["groups"])
    base = {g["name"]: sum(len(h["neurons"]) for h in d["groups"][:i]) for i, g in enumerate(d["groups"])}
    edges = sorted((e[0], e[1], e[3]) for e in d["edges"])
    assert edges == sorted([(base["Input"] + 0, base["Output"] + 0, -1.0), (base["Input"] + 1, base["Output"] + 1, -2.0)])


def test_mapping_section_neuron_range(S, tmp_path):
    # :725-755 `Input.0..2: {core: 0.0}` maps three neurons
    arch = load(S, tmp_path, arch_yaml("a", 1, "tile0", "core0"))
    net = load_net(S, tmp_path, arch, """  name: test
  groups:
    - name: Input
      neurons:
        - 0..2
  edges: []
""", "mappings:\n  - Input.0..2: {core: 0.0}\n")
    d = described(S, arch, net)
    assert [n["core"] for n in d["groups"][0]["neurons"]] == [0, 0, 0]


def test_conv2d_and_dense_hyperedges(S, tmp_path):
    # :773-803 a 3x3x1 input under a 2x2 kernel, stride 1 -> 2x2 outputs, 4 synapses each, weights indexed [y][x][c_in][c_out]
    arch = load(S, tmp_path, arch_yaml("a", 1, "tile0", "core0"))
    net = load_net(S, tmp_path, arch, """  name: test
  groups:
    - name: Input
      neurons:
        - 0..8
    - name: Output
      neurons:
        - 0..3
  edges:
    - Input -> Output:
        type: conv2d
        input_height: 3
        input_width: 3
        input_channels: 1
        kernel_height: 2
        kernel_width: 2
        kernel_count: 1
        stride_height: 1
        stride_width: 1
        weight: [1.0, 2.0, 3.0, 4.0]
""")
    d = described(S, arch, net)
    assert len(d["edges"]) == 16
    base_out = 9
    for oy in range(2):
        for ox in range(2):
            got = sorted((e[0], e[3]) for e in d["edges"] if e[1] == base_out + oy * 2 + ox)
            want = sorted(((oy + ky) * 3 + (ox + kx), [1.0, 2.0, 3.0, 4.0][ky * 2 + kx]) for ky in range(2) for kx in range(2))
            assert got == want
    # :594-638 a hyperedge needs a known type; :805-828 dense attributes must be lists
    for bad in ("    - Input -> Output:\n        weight: [1.0]\n", "    - Input -> Output:\n        type: bogus\n",
                "    - Input -> Output:\n        type: dense\n        weight: 1.0\n"):
        with pytest.raises(Exception):
            load_net(S, tmp_path, arch, "  name: t\n  groups:\n    - name: Input\n      neurons: [0..1]\n    - name: Output\n"
                                        "      neurons: [0..1]\n  edges:\n" + bad)


def test_network_errors(S, tmp_path):
    # :531-592 unknown groups / neuron ids out of range in an edge; :498-529 missing groups / edges sections
    arch = load(S, tmp_path, arch_yaml("a", 1, "tile0", "core0"))
    groups = "  name: t\n  groups:\n    - name: A\n      neurons: [0..1]\n"
    for edges in ("  edges:\n    - B.0 -> A.0: [weight: 1.0]\n", "  edges:\n    - A.0 -> B.0: [weight: 1.0]\n",
                  "  edges:\n    - A.0 -> A.7: [weight: 1.0]\n", "  edges:\n    - A.0 A.1: [weight: 1.0]\n"):
        with pytest.raises(Exception):
            load_net(S, tmp_path, arch, groups + edges)
    with pytest.raises(Exception):
        load_net(S, tmp_path, arch, "  name: t\n  edges: []\n")
    with pytest.raises(Exception):
        load_net(S, tmp_path, arch, groups)
