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
