#!/usr/bin/env python3
"""profiles/<tag>_other_kernels.md from a tools/profile_others.sh run: the kernels the headline step does not contain."""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from parse_rocprof import short  # noqa: E402


def counters(path):
    agg, n = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for f in glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            n[k][r["Counter_Name"]] += 1
    return agg, n


def main():
    tag = sys.argv[1]
    base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_others")
    lines = [f"# Kernels outside the headline step (`{tag}`)", "",
             "`tools/profile_others.sh`: `rocprofv3 --kernel-trace --stats`, then separate `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE` and SQ passes, of "
             "`tools/bench_normals.py` (compute_normals on the 1M-point C3 cloud, k = 30 and radius 0.03) and `tools/bench_radii.py 0.04` (the descriptor "
             "step at radius 0.04: 261 neighbours on average, lists on both sides of 255 points).  HBM traffic per launch = (2 x FETCH_SIZE + WRITE_SIZE) x "
             "1024 B; `vector issue` = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8).", ""]
    for what, title in (("normals", "compute_normals (k-NN, k = 30; radius 0.03)"), ("radii", "descriptor step at radius 0.04")):
        stats = defaultdict(lambda: [0, 0.0])
        p = os.path.join(base, f"{what}_trace", "trace_kernel_stats.csv")
        if not os.path.exists(p):
            fs = glob.glob(os.path.join(base, f"{what}_trace", "**", "*kernel_stats.csv"), recursive=True)
            p = fs[0] if fs else None
        if p is None:
            lines += [f"## {title}", "", "(no trace)", ""]
            continue
        for row in csv.DictReader(open(p)):
            s = stats[short(row["Name"])]
            s[0] += int(row["Calls"])
            s[1] += float(row["TotalDurationNs"])
        fetch, nf = counters(os.path.join(base, f"{what}_fetch"))
        write, nw = counters(os.path.join(base, f"{what}_write"))
        sq, _ = counters(os.path.join(base, f"{what}_sq"))
        lines += [f"## {title}", "", "| kernel | launches | avg us | HBM MB / launch | vector instr. / wave | vector issue | waves / SIMD in flight |", "|---|---|---|---|---|---|---|"]
        for name, (calls, ns) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
            if not name.startswith("k") or ns / calls < 5e3:
                continue
            f = fetch[name]["FETCH_SIZE"] / max(nf[name]["FETCH_SIZE"], 1) if name in fetch else None
            w = write[name]["WRITE_SIZE"] / max(nw[name]["WRITE_SIZE"], 1) if name in write else None
            tb = (2 * f + w) * 1024 / 1e6 if f is not None and w is not None else None
            v = sq.get(name)
            cyc = v["GRBM_GUI_ACTIVE"] / 8.0 if v and v.get("GRBM_GUI_ACTIVE") else 0.0
            issue = v["SQ_ACTIVE_INST_VALU"] * 4 / (1024 * cyc) if cyc else None
            occ = v["SQ_WAVE_CYCLES"] * 4 / (1024 * cyc) if cyc else None
            ipw = v["SQ_INSTS_VALU"] / v["SQ_WAVES"] if v and v.get("SQ_WAVES") else None
            fmt = lambda x, p=2: "" if x is None else f"{x:.{p}f}"
            lines.append(f"| {name} | {calls} | {ns / calls / 1e3:.1f} | {fmt(tb, 1)} | {fmt(ipw, 0)} | {fmt(issue)} | {fmt(occ, 1)} |")
        lines.append("")
        log = os.path.join(base, f"{what}_trace.log")
        if os.path.exists(log):
            lines += ["```"] + [ln.rstrip()[:400] for ln in open(log) if ("ms" in ln and not ln.startswith(("W20", "E20", "I20")))][:12] + ["```", ""]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    open(os.path.join(ROOT, "profiles", f"{tag}_other_kernels.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
