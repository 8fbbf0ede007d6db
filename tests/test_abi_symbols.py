"""The C-ABI libraries must load without a GPU and export every symbol include/*.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header, prefix):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, text)))


@pytest.mark.parametrize("header,lib,prefix", [
    ("sanafe_hip.h", "sana-fe_amd/csrc/libsanafe_hip.so", "sanafe_hip_"),
    ("sanafe_host.h", "sana-fe_amd/host/libsanafe_host.so", "sanafe_"),
])
def test_exports(S, header, lib, prefix):
    S.chip.lib()  # loads both libraries (RTLD_GLOBAL for the device one)
    L = ctypes.CDLL(os.path.join(ROOT, lib))
    names = [n for n in declared(header, prefix) if not (header == "sanafe_host.h" and n.startswith("sanafe_hip_"))]
    assert len(names) >= 10
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, missing


def test_device_count_without_gpu_is_not_an_error(S):
    L = S.chip.hip_lib()
    assert L.sanafe_hip_device_count() >= 0


def test_create_without_device_fails_loudly(S):
    import ctypes as C
    L = S.chip.hip_lib()
    if L.sanafe_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    import nets
    arch, net = nets.example(S)
    chip = S.SpikingChip(arch)
    with pytest.raises(RuntimeError, match="no HIP device"):
        chip.load(net)
