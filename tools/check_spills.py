#!/usr/bin/env python3
"""Register spills per kernel, from the compiler's own listing (no GPU needed): every kernel of csrc/*.hip whose private
(scratch) segment is not empty, with its registers and the occupancy the compiler reports.

A kernel held to a number of waves per SIMD (`amdgpu_waves_per_eu`) that its registers do not fit spills into scratch memory --
correct rows, and in a sweep that waits on `vmcnt(0)` anyway a disaster: K7's full 3-chunk form ran 24 ms instead of 1.5 for a
whole round with 360 bytes of scratch, K6's 4-chunk form 1.09 ms instead of 0.81 with 52.  Run this after touching a kernel or
an occupancy attribute; a few bytes in a cold fallback block (libm atan2 in K6, 12-28 bytes) are harmless.

    python tools/check_spills.py [--min-bytes 1]
"""
from __future__ import annotations

import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "shot_fpfh_amd", "csrc")
FLAGS = "-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I/opt/rocm/include --cuda-device-only -S".split()
EXTRA = {"fpfh.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}  # (as the Makefile)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--min-bytes", type=int, default=1)
    a = ap.parse_args()
    worst = 0
    for f in sorted(x for x in os.listdir(SRC) if x.endswith(".hip")):
        out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, *EXTRA.get(f, []), f, "-o", "-"], cwd=SRC, capture_output=True, text=True)
        if out.returncode:
            print(f, "does not compile:", out.stderr[-500:])
            return 2
        t = out.stdout
        for m in re.finditer(r"^(_Z\S+):\s*; @", t, re.M):
            seg = t[m.end():m.end() + 800000]
            sc = re.search(r"; ScratchSize: (\d+)", seg)
            if not sc or int(sc.group(1)) < a.min_bytes or "rocprim" in m.group(1):
                continue
            name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            name = re.sub(r"\(anonymous namespace\)::", "", name)
            name = re.sub(r"\(.*$", "", name).replace("void ", "")
            vg = re.search(r"; NumVgprs: (\d+)", seg).group(1)
            oc = re.search(r"; Occupancy: (\d+)", seg).group(1)
            print(f"{f:18s} {name:60s} scratch {int(sc.group(1)):4d} B  vgpr {vg:>3s}  waves/SIMD {oc}")
            worst = max(worst, int(sc.group(1)))
    print("largest private segment:", worst, "bytes")
    return 0


if __name__ == "__main__":
    sys.exit(main())
