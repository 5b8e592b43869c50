#!/usr/bin/env python3
"""tools/bench_radii.py + tools/bench_nbins.py outputs -> profiles/<tag>_radius_sweep.md.
    python tools/radius_sweep_md.py <radii.txt> <nbins.txt> <tag>"""
import ast
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
radii, nbins, tag = sys.argv[1], sys.argv[2], sys.argv[3]
from shot_fpfh_amd import _ffi

try:
    build = _ffi.load().sf_version().decode()
except Exception:  # noqa: BLE001
    build = "?"
out = [f"# The bench step at other radii and bin counts ({tag}, build {build}, one MI355X)", "",
       "`python tools/bench_radii.py` -- FPFH(5 bins) + SHOT for all points of the 1M-point uniform cloud, 5 timed steps (every step "
       "repeats the search of the same range: planned from the record, no read-back):", "",
       "| radius | mean neighbours | ms / step | ns per (keypoint, neighbour) pair | K2 | K6 (+ tail) | K7 (+ tail) | K5 (+ tail) |", "|---|---|---|---|---|---|---|---|"]


def pair(k, main, *tails):
    t = sum(k.get(x, 0.0) for x in tails)
    return f"{k.get(main, 0.0):.2f}" + (f" + {t:.2f}" if t > 0.005 else "")


for ln in open(radii):
    m = re.match(r"r=([\d.]+) kbar=(\d+) step ([\d.]+) ms\s+([\d.]+) ns/pair\s+(\{.*\})", ln.strip())
    if not m:
        continue
    k = ast.literal_eval(m.group(5))
    k2 = sum(v for n, v in k.items() if n.startswith("k2_"))
    out.append(f"| {m.group(1)} | {m.group(2)} | {float(m.group(3)):.2f} | {m.group(4)} | {k2:.2f} | {pair(k, 'k6_spfh', 'k6_spfh_tail', 'k6_spfh_mid')} | "
               f"{pair(k, 'k7_fpfh', 'k7_fpfh_tail', 'k7_fpfh_mid')} | {pair(k, 'k5_shot', 'k5_shot_tail', 'k5_shot_tail_stream', 'k5_shot_mid')} |")
out += ["", "`python tools/bench_nbins.py` -- FPFH alone, radius 0.03, by bin count (ms per step; K6 + K7):", "",
        "| n_bins | bins | ms / step | K6 (+ pack) | K7 |", "|---|---|---|---|---|"]
for ln in open(nbins):
    m = re.match(r"(\d+) ([\d.]+) (\{.*\})", ln.strip())
    if not m:
        continue
    k = ast.literal_eval(m.group(3))
    nb = int(m.group(1))
    out.append(f"| {nb} | {nb ** 3} | {float(m.group(2)):.2f} | {pair(k, 'k6_spfh', 'k6_spfh_pack')} | {k.get('k7_fpfh', 0.0):.2f} |")
out.append("")
open(os.path.join(ROOT, "profiles", f"{tag}_radius_sweep.md"), "w").write("\n".join(out))
print("\n".join(out))
