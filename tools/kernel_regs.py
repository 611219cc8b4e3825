#!/usr/bin/env python
"""VGPR / spill / scratch / LDS of every kernel of one translation unit (cross-compiles, no GPU needed).

    python tools/kernel_regs.py scan_packed.hip -DTPQ_PACKED_M=64 [--grep scan_packed_kernel]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    args = sys.argv[1:]
    pat = None
    if "--grep" in args:
        i = args.index("--grep")
        pat = args[i + 1]
        del args[i:i + 2]
    src, flags = args[0], args[1:]
    with tempfile.TemporaryDirectory() as td:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
                               "-ffp-contract=off", "-fno-fast-math", "-DNDEBUG", *flags, "-x", "hip", "-c",
                               os.path.join(ROOT, "torchpq_amd", "csrc", src), "-o", os.path.join(td, "o.o"),
                               "-save-temps"], cwd=td, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(td) if f.endswith(".s") and "amdgcn" in f][0]
        text = open(os.path.join(td, asm)).read()
    for b in text.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        if pat and pat not in dn:
            continue
        f = lambda k: re.search(rf"\.{k}:\s+(\d+)", b).group(1)
        print(f"{dn[:90]:90s} vgpr {f('vgpr_count'):>3s} agpr {b.split()[0]:>3s} spill {f('vgpr_spill_count'):>3s} "
              f"scratch {f('private_segment_fixed_size'):>4s} sgpr {f('sgpr_count'):>3s}")


if __name__ == "__main__":
    main()
