"""Guards the one property of the generated code the delivery kernel's speed hangs on ("loads that can
be counted"): inside the stream loops every wait for a synapse-word load is a counted one (`s_waitcnt vmcnt(n)`, n =
groups in flight - 1), never `vmcnt(0)`.  A refill load behind a branch, or loads reordered by the scheduler, silently
turn them into vmcnt(0) -- one group per wave in flight instead of four, 15 % slower -- without failing any parity test."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sana-fe_amd", "csrc", "sanafe_hip.hip")


@pytest.fixture(scope="module")
def isa(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = tmp_path_factory.mktemp("isa") / "kernels.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-S",
                    "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-o", str(out), SRC], check=True,
                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out.read_text().split("\n")


def _function(lines, mangled_part):
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and ":" in l and mangled_part in l.split(":")[0])
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
    return lines[start:end]


def _steady_state_waits(fn, depth_in_flight):
    """vmcnt values waited for between the first refill of the stream (the load after the `depth_in_flight` initial ones)
    and the last one: the unrolled body of the stream loop."""
    nt = [i for i, l in enumerate(fn) if "global_load_dwordx4" in l and l.rstrip().endswith(" nt")]
    assert len(nt) >= 2 * depth_in_flight, "expected %d initial and %d refill loads, found %d non-temporal loads" % (depth_in_flight, depth_in_flight, len(nt))
    region = fn[nt[depth_in_flight]:nt[-1]]
    return [int(x) for l in region for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", l)]


@pytest.mark.parametrize("fmt,depth_in_flight,bitmap", [(7, 4, 0), (7, 3, 1), (6, 4, 0), (0, 4, 0), (3, 4, 0), (4, 2, 0)])
def test_stream_loads_are_counted(isa, fmt, depth_in_flight, bitmap):
    fn = _function(isa, "deliver_kernelILi%dELb0ELb0ELb0ELi256ELb%dELb0E" % (fmt, bitmap))
    waits = _steady_state_waits(fn, depth_in_flight)
    assert waits, "the stream loop of deliver_kernel<%d> waits for no load at all?" % fmt
    assert min(waits) >= depth_in_flight - 1, (fmt, waits)


def _metadata(isa):
    """name -> (vgprs, scratch bytes) of every kernel, from the code object's metadata at the end of the assembly."""
    out = {}
    for block in "\n".join(isa).split("  - .agpr_count:")[1:]:
        def field(name):
            return re.search(r"\.%s:\s+(\S+)" % name, block).group(1)
        out[field("name")] = (int(field("vgpr_count")), int(field("private_segment_fixed_size")))
    return out


def test_hot_kernels_keep_their_occupancy_and_do_not_spill(isa):
    """The measured operating points: the neuron kernels fit 64 VGPRs (8 wavefronts per SIMD), the
    bitmap-record delivery kernels 80 (6 per SIMD), every other delivery instantiation 96 (5 per SIMD) -- and the headline
    instantiations use no scratch at all (a spilled dword in the stream loop costs ~10 % of the launch)."""
    meta = _metadata(isa)
    neuron = {k: v for k, v in meta.items() if "neuron_kernel" in k}
    assert len(neuron) == 5
    for name, (vgprs, scratch) in neuron.items():
        assert vgprs <= 64 and scratch == 0, (name, vgprs, scratch)
    deliver = {k: v for k, v in meta.items() if "deliver_kernelILi" in k and "event_deliver" not in k}
    assert len(deliver) >= 36  # every instantiation launch_deliver can pick (deliver_variants)
    for name, (vgprs, scratch) in deliver.items():
        assert vgprs <= 96 and scratch <= 16, (name, vgprs, scratch)
    # format 7 on bitmap records, 256 threads, with and without sub-accumulators (the PUSH prologue is gone: the host decides)
    hot = {k: v for k, v in deliver.items() if "deliver_kernelILi7ELb0ELb0ELb0ELi256ELb1E" in k}
    assert len(hot) == 2
    for name, (vgprs, scratch) in hot.items():
        assert vgprs <= 80 and scratch == 0, (name, vgprs, scratch)
    # event-driven delivery: (4, 8 lanes x 1 unit, 4 lanes x 2 units) per block x 4 / 5 code bits x 4 / 8 / 16 wavefronts, 64 registers (8 wavefronts per
    # SIMD: two 16-wavefront workgroups per CU), no scratch
    event = {k: v for k, v in meta.items() if "event_deliver_kernel" in k}
    assert len(event) == 36  # (x 2: the group-major and the neuron-major block table)
    for name, (vgprs, scratch) in event.items():
        assert vgprs <= 64 and scratch == 0, (name, vgprs, scratch)
