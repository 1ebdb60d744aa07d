#!/usr/bin/env python3
"""Registers, scratch, LDS and occupancy of every kernel, from the compiler's own report.

  hipcc ... -c engine.hip -Rpass-analysis=kernel-resource-usage 2> report.txt
  python tools/kernel_resources.py report.txt [substring]
"""
import re
import subprocess
import sys


def main():
    txt = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
        name = b.split("\n")[0].split(" [-R")[0].strip()
        try:
            name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
        name = re.sub(r"\(anonymous namespace\)::|msm::", "", name)
        name = re.sub(r"\(.*\)$", "", name)
        if want not in name:
            continue
        g = lambda k: int(re.search(k + r": (\d+)", b).group(1))  # noqa: E731
        scratch, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
        print(f"{name[:70]:70s} VGPR {g('VGPRs'):3d} AGPR {g('AGPRs'):3d} scratch {scratch:4d} occ {occ} LDS {lds}")


if __name__ == "__main__":
    main()
