"""The programmatic presets must equal what the YAML front-end reads from the reference's
architecture / SNN descriptions (checked only where /root/reference is present)."""
import nets  # noqa: E402
import pytest

from conftest import REFERENCE, have_reference

pytestmark = pytest.mark.skipif(not have_reference(), reason="reference not present")


def arch_view(S, arch, n_tiles=2):
    # compare a slice of the (identical) tiles unit for unit; counts are compared separately
    ncores = sum(len(t.cores) for t in arch.tiles[:n_tiles])
    arch.tiles = arch.tiles[:n_tiles]
    arch._cores = arch._cores[:ncores]
    d = S.description.describe(S.to_desc(arch, S.description.Network()))
    return {k: d[k] for k in ("noc", "sync", "tiles", "cores")}


@pytest.mark.parametrize("preset,yaml_name,kw", [
    ("example_chip", "example_chip.yaml", {}),
    ("loihi", "loihi.yaml", {}),
    ("truenorth", "truenorth.yaml", {"n_tiles": 4096}),
])
def test_preset_equals_yaml(S, preset, yaml_name, kw):
    pa, ya = getattr(S.presets, preset)(api=S.description, **kw), S.yaml_io.load_arch(REFERENCE + "/arch/" + yaml_name)
    assert (len(pa.tiles), pa.core_count) == (len(ya.tiles), ya.core_count)
    a, b = arch_view(S, pa), arch_view(S, ya)
    # unit attribute dicts from YAML additionally carry the keys the parser forwards verbatim
    for ca, cb in zip(a["cores"], b["cores"]):
        for ua, ub in zip(ca["units"], cb["units"]):
            for k in list(ub["attrs"]):
                if k not in ua["attrs"] and k in ("update_every_timestep", "log_energy", "log_latency"):
                    del ub["attrs"][k]
    assert a == b


def test_loihi_large_preset_sample(S):
    # 1024 tiles x 4 cores x 1031 units is large: compare a 2-tile slice, unit for unit
    y = S.yaml_io.load_arch(REFERENCE + "/arch/loihi_large.yaml")
    assert (y.noc_width, y.noc_height, len(y.tiles), y.core_count) == (256, 128, 1024, 4096)
    a, b = arch_view(S, S.presets.loihi_large(n_tiles=2, api=S.description)), arch_view(S, y)
    assert a == b


def test_example_snn_preset(S):
    arch = S.presets.example_chip(api=S.description)
    a = S.description.describe(S.to_desc(arch, nets.example_snn(S, arch)))
    arch2 = S.yaml_io.load_arch(REFERENCE + "/arch/example_chip.yaml")
    b = S.description.describe(S.to_desc(arch2, S.yaml_io.load_net(REFERENCE + "/snn/example_snn.yaml", arch2)))
    assert a["groups"] == b["groups"]
    assert a["edges"] == b["edges"]
