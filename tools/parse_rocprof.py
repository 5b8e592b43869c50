#!/usr/bin/env python3
"""Summarise a tools/profile_gpu.sh run (rocprofv3 CSVs under gpurun_out/prof_<tag>/) into
profiles/<tag>_summary.md, profiles/<tag>_traffic.json (HBM bytes per launch of each hot kernel) and
profiles/<tag>_sustained_clock.json (shader clock per kernel: GRBM_GUI_ACTIVE / 8 XCDs / dispatch duration), both stamped
with the library's build id so that bench.py never quotes them for another build.

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): FETCH_SIZE and
WRITE_SIZE are collected in SEPARATE --pmc passes, are reported in KiB, and on gfx950 FETCH_SIZE counts
128-B requests as 64 B for wide coalesced reads, so the read side is doubled:
    traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes.
"""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHORT = [
    ("k_rows_gather", "k8_rows_gather"), ("k_rows_nonzero", "k8_rows_nonzero"), ("k_icp_", "i1_icp"), ("k_transform", "i0_transform"),
    ("k_voxel", "v_voxel"), ("k_azimuth", "k5_azimuth_idx"), ("k_spfh_generic", "k6_spfh_generic"), ("k_spfh_pack", "k6_spfh_pack"), ("k_spfh_repack", "k6_spfh_pack"), ("k_fpfh_generic", "k7_fpfh_generic"),
    ("k_lrf_from_cov", "k4_lrf_from_cov"), ("k_shot_lrf", "k4_shot_lrf"), ("k_lrf_", "k4_shot_lrf"), ("k_shot", "k5_shot"), ("k_fpfh", "k7_fpfh"), ("k_spfh_export", "k6_spfh_export"),
    ("k_spfh", "k6_spfh"), ("k_radius<2>", "k2_radius_slots"), ("k_radius<1>", "k2_radius_fill"), ("k_radius<0>", "k2_radius_count"),
    ("k_radius<true>", "k2_radius_fill"), ("k_radius<false>", "k2_radius_count"), ("k_export_lists", "k2_export_lists"),
    ("k_radius", "k2_radius"), ("k_normals_max", "k1_normals_max"), ("k_pca_cov", "k3_normals"), ("k_pca_solve", "k3_normals"), ("k_normals", "k3_normals"), ("k_match_tile", "k8_match_tile"), ("k_ransac", "k9_ransac_score"),
    ("k_match_half", "k8_match_half"), ("k_half_convert", "k8_half_convert"), ("k_half_final", "k8_half_final"),
    ("k_half_window", "k8_half_window"), ("k_half_max", "k8_half_max"), ("k_match_gemm", "k8_match_gemm"),
    ("k_match_decide", "k8_match_decide"), ("k_row_sqnorm", "k8_row_sqnorm"),
    ("k_gather_sorted", "k1_gather_sorted"), ("k_gather_normals", "k1_gather_normals"), ("k_count_stats", "k2_reduce"),
    ("k_layer_hist", "k1_layer_hist"), ("k_select_slab", "k1_select_slab"), ("k_extract_z", "k1_extract_z"), ("k_col_candidates", "k8_col_candidates"), ("k_lrf_eigen", "k4_lrf_eigen"), ("k_pca", "k3_pca"), ("k_cell_ids", "k1_cell_ids"), ("k_cell_start", "k1_cell_start"),
    ("k_bbox", "k1_bbox"), ("radix_sort", "rocprim_radix_sort"), ("merge_sort", "rocprim_radix_sort"),
    ("scan", "rocprim_scan"), ("reduce", "rocprim_reduce"), ("copyBuffer", "hip_copy"), ("fillBuffer", "hip_fill"),
]


# round 4: second launches and selections (checked before the table above; template arguments as rocprofv3 prints them)
SHORT_RE = [
    (r"k_i8_min", "k8_match_i8"), (r"k_i8_collect", "k8_i8_collect"), (r"k_i8_final", "k8_i8_final"), (r"k_i8_live|k_i8_place", "k8_i8_live"),
    (r"k_i8_convert|k_i8_rowstat", "k8_i8_convert"), (r"k_i8_max", "k8_i8_max"), (r"k_i8_window", "k8_i8_window"),
    (r"k_i8_gather_rows|k_half_gather_rows", "k8_gather_rows"), (r"k_i8_scatter|k_half_scatter", "k8_scatter_results"),
    (r"radix_sort", "rocprim_radix_sort"), (r"k_radius<2, ", "k2_radius_slots"), (r"k_radius<1, true>", "k2_radius_refill"), (r"k_radius<1, false>", "k2_radius_fill"),
    (r"k_radius<0, true>", "k2_sample"), (r"k_radius<0, false>", "k2_radius_count"), (r"k_iota_stride", "k2_sample"),
    (r"k_patch_offsets", "k2_select"), (r"select|partition", "rocprim_select"), (r"k_shot_long|k_shot_team", "k5_shot_tail"), (r"k_fpfh_mcl", "k7_fpfh_tail"),
    (r"k_fpfh_tail", "k7_fpfh_tail"), (r"k_spfh<[^>]*, true>", "k6_spfh_tail"), (r"k_pca_cov<0, true>", "k3_normals_tail"),
    (r"k_cell_fill_long", "k1_cell_start"), (r"k_cell_count", "k1_cell_count"), (r"k_cell_scan", "k1_cell_scan"), (r"k_cell_place", "k1_cell_place"),
    (r"k_cell_settle", "k1_cell_settle"),
]


def short(name):
    for pat, s in SHORT_RE:
        if re.search(pat, name):
            return s
    for pat, s in SHORT:
        if pat in name:
            return s
    return re.sub(r"\(.*", "", name)[:40]


def read_counter(path, counter):
    agg = defaultdict(list)
    if not os.path.exists(path):
        return agg
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] == counter:
            agg[short(row["Kernel_Name"])].append(float(row["Counter_Value"]))
    return agg


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    cmd = sys.argv[2] if len(sys.argv) > 2 else ("python3 bench.py --no-cpu-baseline --no-dropin --no-parity --no-density --no-defaults "
                                                 "--no-match --no-normals (20 timed steps + the >= 2 s sustained window)")
    bench_run = len(sys.argv) <= 2  # only the bench profile feeds bench.py's roofline.traffic
    base = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    stats = defaultdict(lambda: [0, 0.0])
    for row in csv.DictReader(open(os.path.join(base, "trace", "trace_kernel_stats.csv"))):
        s = stats[short(row["Name"])]
        s[0] += int(row["Calls"])
        s[1] += float(row["TotalDurationNs"])
    total = sum(v[1] for v in stats.values())
    fetch = read_counter(os.path.join(base, "pmc_fetch", "pmc_counter_collection.csv"), "FETCH_SIZE")
    write = read_counter(os.path.join(base, "pmc_write", "pmc_counter_collection.csv"), "WRITE_SIZE")
    hit = read_counter(os.path.join(base, "pmc_l2", "pmc_counter_collection.csv"), "TCC_HIT_sum")
    miss = read_counter(os.path.join(base, "pmc_l2", "pmc_counter_collection.csv"), "TCC_MISS_sum")
    lines = [f"# rocprofv3 summary `{tag}`", "",
             f"Command: `rocprofv3 --kernel-trace --stats -- {cmd}` "
             "(+ separate `--pmc FETCH_SIZE`, `--pmc WRITE_SIZE`, `--pmc TCC_HIT_sum TCC_MISS_sum` passes).", "",
             "HBM traffic/launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md).", "",
             "| kernel | calls | avg us | % of GPU time | FETCH_SIZE KiB/launch (raw) | WRITE_SIZE KiB/launch | HBM MB/launch (corrected) | L2 hit rate |",
             "|---|---|---|---|---|---|---|---|"]
    traffic = {}
    for name, (calls, ns) in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        f = sum(fetch[name]) / len(fetch[name]) if fetch.get(name) else None
        w = sum(write[name]) / len(write[name]) if write.get(name) else None
        hr = None
        if hit.get(name) and miss.get(name):
            h, m = sum(hit[name]), sum(miss[name])
            hr = h / (h + m) if h + m else None
        tb = (2 * f + w) * 1024 if f is not None and w is not None else None
        if tb is not None and name.startswith("k"):
            traffic[name] = tb
        lines.append(f"| {name} | {calls} | {ns / calls / 1e3:.1f} | {100 * ns / total:.2f} | "
                     f"{'' if f is None else f'{f:.0f}'} | {'' if w is None else f'{w:.0f}'} | "
                     f"{'' if tb is None else f'{tb / 1e6:.1f}'} | {'' if hr is None else f'{hr:.3f}'} |")
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    open(os.path.join(ROOT, "profiles", f"{tag}_summary.md"), "w").write("\n".join(lines) + "\n")
    if bench_run:
        # stamped with the build id of the library the counters were taken on: bench.py quotes the file for that build only
        sys.path.insert(0, ROOT)
        from shot_fpfh_amd import _ffi

        traffic["_build"] = _ffi.load().sf_version().decode()
        traffic["_source"] = f"profiles/{tag}_summary.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `{cmd}`)"
        json.dump(traffic, open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json"), "w"), indent=1)
        # shader clock during every kernel of the run (the sustained window included): GRBM_GUI_ACTIVE is summed over the 8 XCDs
        cpath = os.path.join(base, "pmc_clock", "pmc_counter_collection.csv")
        if os.path.exists(cpath):
            per = defaultdict(lambda: [0.0, 0.0, 0])  # kernel -> [cycles, ns, dispatches]
            for row in csv.DictReader(open(cpath)):
                if row["Counter_Name"] != "GRBM_GUI_ACTIVE":
                    continue
                d = per[short(row["Kernel_Name"])]
                d[0] += float(row["Counter_Value"])
                d[1] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
                d[2] += 1
            clk = {k: {"clock_mhz": round(v[0] / 8.0 / v[1] * 1e3, 1), "dispatches": v[2], "avg_us": round(v[1] / v[2] / 1e3, 1)}
                   for k, v in per.items() if v[1] > 0 and k.startswith("k")}
            hot = [k for k in ("k5_shot", "k6_spfh", "k7_fpfh", "k2_radius_slots") if k in per]
            tot_c, tot_t = sum(per[k][0] for k in hot), sum(per[k][1] for k in hot)
            rec = {"_build": traffic["_build"],
                   "_source": f"profiles/{tag}_sustained_clock.json: rocprofv3 --pmc GRBM_GUI_ACTIVE pass of `{cmd}` (the run that contains "
                              "the sustained window; under counter collection the dispatches are serialised)",
                   "clock_mhz_time_weighted_hot_kernels": round(tot_c / 8.0 / tot_t * 1e3, 1) if tot_t else None,
                   "kernels": clk}
            json.dump(rec, open(os.path.join(ROOT, "profiles", f"{tag}_sustained_clock.json"), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
