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


def test_malformed_images_are_rejected_with_their_reason(S):
    """sanafe_hip_chip_create checks every index the kernels will dereference BEFORE touching a device, so a bad image
    fails with its reason here as on a GPU box (a faulting kernel can take a whole node down)."""
    import ctypes as C
    import nets
    L, H = S.chip.lib(), S.chip.hip_lib()
    arch, net = nets.random_loihi(S, n_tiles=1, neurons_per_core=70, out_degree=12, arch_kind="loihi")
    built = S.chip._Lowered(arch, net)
    h = C.c_void_p()
    assert L.sanafe_chip_create(built.address, -1, 1, 0, C.byref(h)) == 0
    try:
        im = S.chip.HipImage()
        L.sanafe_chip_get_image.argtypes = [C.c_void_p, C.POINTER(S.chip.HipImage)]
        assert L.sanafe_chip_get_image(h, C.byref(im)) == 0
        H.sanafe_hip_chip_create.argtypes = [C.POINTER(S.chip.HipImage), C.c_int, C.POINTER(C.c_void_p)]
        H.sanafe_hip_chip_destroy.argtypes = [C.c_void_p]

        def create():
            out = C.c_void_p()
            rc = H.sanafe_hip_chip_create(C.byref(im), 0, C.byref(out))
            msg = H.sanafe_hip_last_error().decode() if rc != 0 else ""
            if rc == 0:
                H.sanafe_hip_chip_destroy(out)
            return rc, msg

        rc, msg = create()  # the untouched image is accepted; without a GPU the only complaint is the missing device
        assert rc == 0 or "no HIP device" in msg, msg

        def broken(field, value, expect):
            saved = getattr(im, field)
            setattr(im, field, value)
            rc, msg = create()
            setattr(im, field, saved)
            assert rc != 0 and expect in msg, (field, msg)

        broken("n_slots", im.n_slots + 1, "multiple of 64")
        broken("ring_slots", 0, "ring_slots")
        broken("n_cost_classes", 0, "cost classes")
        broken("slot_offset", 32, "global slot window")

        def broken_entry(array, index, value, expect):
            saved = array[index]
            array[index] = value
            rc, msg = create()
            array[index] = saved
            assert rc != 0 and expect in msg, msg

        broken_entry(im.ax_pre, 0, 0xfffffff0, "bad pre slot")
        broken_entry(im.core_nbase, 1, im.core_nbase[1] + 1, "64-aligned")
        broken_entry(im.syn_meta, 0, (im.syn_meta[0] & ~0xffff) | 0xffff, "post neuron outside its core")
        broken_entry(im.slice_axon_end, 0, im.n_axons + 5, "bad axon range")
        broken_entry(im.slot_cls, 0, (im.slot_cls[0] & ~(1023 << 6)) | (1023 << 6), "bad cost class")
    finally:
        L.sanafe_chip_destroy(h)
