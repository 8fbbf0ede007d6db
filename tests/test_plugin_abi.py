"""The plugin ABI headers (sana-fe_amd/host/plugin_abi) must be source-compatible with the reference's
plugin surface: its own example plugin recompiles unchanged against them and exports the factory."""
import os
import subprocess

import pytest

from conftest import REFERENCE, ROOT, have_reference


@pytest.mark.skipif(not have_reference(), reason="reference not present")
def test_reference_plugin_recompiles_unchanged(tmp_path):
    so = tmp_path / "libhh_ref_src.so"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-Wall",
                           "-I", os.path.join(ROOT, "sana-fe_amd", "host", "plugin_abi"),
                           os.path.join(REFERENCE, "plugins", "hodgkin_huxley.cpp"), "-o", str(so)])
    syms = subprocess.check_output(["nm", "-D", str(so)]).decode()
    assert " T create_hodgkin_huxley" in syms


def test_in_tree_plugin_exports_factory():
    so = os.path.join(ROOT, "tests", "plugins", "libhodgkin_huxley.so")
    assert os.path.exists(so), "run `make -C sana-fe_amd`"
    assert " T create_hodgkin_huxley" in subprocess.check_output(["nm", "-D", so]).decode()


def test_plugin_network_maps_with_host_slots(S):
    import nets
    arch, net = nets.hodgkin_huxley(S)
    im, slot_of = S.map_only(arch, net)
    models = im["slot_cls"][slot_of] & 7
    assert (models[:12] == 4).all()          # SANAFE_SOMA_HOST
    assert (models[12:] == 1).all()          # LIF
