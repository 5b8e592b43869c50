#!/usr/bin/env python3
"""profiles/<tag>_match_summary.md from a tools/profile_match_r6.sh run: per K8 kernel the trace's average duration, HBM traffic
(FETCH_SIZE x 2 + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md), L2 hit rate, and from the SQ passes the share of the
kernel's time its matrix cores are busy, its vector issue, LDS activity and the clock the chip held."""
import csv
import glob
import os
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from parse_rocprof import short  # noqa: E402


PASSES = 2  # run_config4.py runs K8 twice (one untimed pass, one with the launch timers on)


def counters(path):
    """per kernel: counter sums over all dispatches, and the summed dispatch time (ns) of the dispatches that carry GRBM_GUI_ACTIVE"""
    agg, dur = defaultdict(lambda: defaultdict(float)), defaultdict(float)
    files = glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE" and "End_Timestamp" in r:
                dur[k] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    return agg, dur


def main():
    tag, n = sys.argv[1], int(sys.argv[2])
    base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}_match")
    stats = defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(os.path.join(base, "trace", "trace_kernel_stats.csv"))):
        s = stats[short(row["Name"])]
        s[0] += int(row["Calls"])
        s[1] += float(row["TotalDurationNs"])
    fetch, _ = counters(os.path.join(base, "pmc_fetch"))
    write, _ = counters(os.path.join(base, "pmc_write"))
    l2, _ = counters(os.path.join(base, "pmc_l2"))
    sa, da = counters(os.path.join(base, "sq_a"))
    sb, db = counters(os.path.join(base, "sq_b"))
    k8 = {k: v for k, v in stats.items() if k.startswith("k8") or k.startswith("k9")}
    total = sum(v[1] for v in k8.values())
    ops = 2.0 * n * n * 352
    lines = [f"# K8 at BASELINE config 4's size under rocprofv3 (`{tag}`)", "",
             f"`tools/profile_match_r6.sh {tag} {n}`: `rocprofv3 --kernel-trace --stats -- python3 tools/run_config4.py {n} 2000` (two {n}-point clouds' SHOT rows; "
             f"the script runs K8 {PASSES} times -- one untimed pass, one with the launch timers on -- and every figure below is PER PASS: a kernel's launches of "
             "one pass added up, the pilot slab's launch of the integer pre-filter included), then SEPARATE `--pmc` passes of the same command: FETCH_SIZE; WRITE_SIZE; "
             "TCC_HIT_sum TCC_MISS_sum; SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE; "
             "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE.", "",
             "Fabric traffic per pass = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B (the gfx950 correction of MI355X_MICROARCH.md; reads served by the Infinity Cache "
             "count).  `vector issue` = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); `LDS` likewise from SQ_ACTIVE_INST_LDS; "
             "`MFMA busy` = SQ_VALU_MFMA_BUSY_CYCLES / (1024 x GRBM_GUI_ACTIVE / 8) -- this counter is NOT in the 4-cycle unit of the SQ_ACTIVE family: "
             "tools/ubench/mfma_rates under the same counters (profiles/" + tag + "_mfma_counter_calibration.txt) gives the value a bare back-to-back loop reads; "
             "`clock` = GRBM_GUI_ACTIVE / 8 / the dispatches' duration under counter collection.", "",
             "| kernel | launches / pass | ms / pass (trace) | % of K8+K9 time | fabric GB / pass | L2 hit rate | MFMA busy | vector issue | LDS | LDS conflict cycles / LDS cycles | clock MHz | rate of the pass |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|"]
    fmt = lambda x, p=3: "" if x is None else f"{x:.{p}f}"
    for name, (calls, ns) in sorted(k8.items(), key=lambda kv: -kv[1][1]):
        f = fetch[name].get("FETCH_SIZE") if name in fetch else None
        w = write[name].get("WRITE_SIZE") if name in write else None
        tb = (2 * f + w) * 1024 / PASSES if f is not None and w is not None else None
        h, m = l2[name].get("TCC_HIT_sum", 0.0), l2[name].get("TCC_MISS_sum", 0.0)
        hr = h / (h + m) if h + m else None
        cyc_a = sa[name].get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        cyc_b = sb[name].get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        mfma = sb[name].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024 * cyc_b) if cyc_b else None
        valu = sa[name].get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (1024 * cyc_a) if cyc_a else None
        lds = sb[name].get("SQ_ACTIVE_INST_LDS", 0.0) * 4 / (1024 * cyc_b) if cyc_b else None
        conf = sb[name].get("SQ_LDS_BANK_CONFLICT", 0.0) / sb[name]["SQ_ACTIVE_INST_LDS"] if sb[name].get("SQ_ACTIVE_INST_LDS") else None
        clk = cyc_b / db[name] * 1e3 if db.get(name) else None
        rate = ""
        if name == "k8_match_i8":
            rate = f"{ops / (ns / PASSES * 1e-9) / 1e15:.2f} Pop/s (2 m1 m2 d = {ops:.3g} op)"
        lines.append(f"| {name} | {calls / PASSES:g} | {ns / PASSES / 1e6:.3f} | {100 * ns / total:.1f} | {'' if tb is None else f'{tb / 1e9:.2f}'} | {fmt(hr)} | {fmt(mfma, 2)} | "
                     f"{fmt(valu, 2)} | {fmt(lds, 2)} | {fmt(conf, 2)} | {fmt(clk, 0)} | {rate} |")
    lines += ["", f"All K8 + K9 kernels: {total / PASSES / 1e6:.1f} ms per pass.", "",
              "Wall time of the K8 call and the rest of config 4 (the traced run's own print-out; tracing adds a little):", "", "```"]
    for ln in open(os.path.join(base, "trace.log")):
        if ln.startswith(("config 4", "  ")) and "kernels (launches" not in ln:
            lines.append(ln.rstrip())
    lines.append("```")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    open(os.path.join(ROOT, "profiles", f"{tag}_match_summary.md"), "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
