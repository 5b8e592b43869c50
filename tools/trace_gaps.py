#!/usr/bin/env python3
"""Busy / idle time of the GPU over the last steps of a rocprofv3 --kernel-trace CSV: per kernel name the mean duration and
the mean idle gap in FRONT of it (start - end of the previous kernel, 0 if they overlap).

    python tools/trace_gaps.py <kernel_trace.csv> <kernels per step | auto> [steps]

auto: a step starts with K1's first kernel (k_cell_count / k_cell_ids): the last `steps` + 1 of them delimit the steps.
Also prints the largest single idle gap inside those steps (round 5: a repeated step has no host read-back left).
"""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
if sys.argv[2] == "auto":
    starts = [i for i, r in enumerate(rows) if "k_cell_count" in r[2] or "k_cell_ids" in r[2]]
    starts = starts[-(steps + 1):]
    rows = rows[starts[0]:starts[-1]]
    steps = len(starts) - 1
else:
    per_step = int(sys.argv[2])
    rows = rows[-per_step * steps:]
dur, gap, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
prev_end = rows[0][0]
for s, e, name in rows:
    short = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    short = re.sub(r"\(.*$", "", short)[:70]
    dur[short] += e - s
    gap[short] += max(0, s - prev_end)
    cnt[short] += 1
    prev_end = max(prev_end, e)
span = prev_end - rows[0][0]
big, pe = (0, ""), rows[0][0]
for s_, e_, name in rows:
    if s_ - pe > big[0]:
        big = (s_ - pe, re.sub(r"\(.*$", "", re.sub(r"^void ", "", name).replace("(anonymous namespace)::", ""))[:60])
    pe = max(pe, e_)
print(f"largest idle gap inside the steps: {big[0] / 1e3:.2f} us, in front of {big[1]}")
print(f"span per step {span / steps / 1e3:.1f} us; busy {sum(dur.values()) / steps / 1e3:.1f} us; idle {sum(gap.values()) / steps / 1e3:.1f} us")
for k in sorted(dur, key=lambda k: -dur[k] - gap[k]):
    print(f"{dur[k] / steps / 1e3:9.2f} us  gap in front {gap[k] / steps / 1e3:7.2f} us  x{cnt[k] / steps:4.1f}  {k}")
