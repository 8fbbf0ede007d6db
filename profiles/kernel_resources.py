#!/usr/bin/env python3
"""Registers, scratch and LDS of every kernel instantiation, read from the gfx950 code object's metadata (no GPU needed):
    python3 profiles/kernel_resources.py > profiles/rNN_kernel_resources.txt
What to look for: `scratch` / `spill` must be 0 on the hot instantiations (a spilled dword in the stream loop of
deliver_kernel costs ~10 % of the launch), and `vgpr` <= 96 keeps five wavefronts per SIMD (512 / 96)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "sana-fe_amd", "csrc", "sanafe_hip.hip")


def main():
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "kernels.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-munsafe-fp-atomics", "-S",
                        "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-o", out, SRC] + sys.argv[1:], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    rows = []
    for b in txt.split("  - .agpr_count:")[1:]:
        def field(name):
            m = re.search(r"\.%s:\s+(\S+)" % name, b)
            return m.group(1) if m else "?"
        name = subprocess.run(["c++filt", field("name")], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(.*", "", name).replace("void ", "")
        rows.append((name, field("vgpr_count"), field("sgpr_count"), field("private_segment_fixed_size"), field("vgpr_spill_count"),
                     field("group_segment_fixed_size")))
    print("%-62s %5s %5s %8s %6s %10s" % ("kernel", "vgpr", "sgpr", "scratch", "spill", "static LDS"))
    for r in sorted(rows):
        print("%-62s %5s %5s %8s %6s %10s" % r)


if __name__ == "__main__":
    main()
