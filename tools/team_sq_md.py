"""profiles/<tag>_k5_team.md from the counters of tools/pmc_team.sh: what K5's team form (k_shot_team) and the register-cached form
beside it spend per keypoint at a radius whose lists exceed 255 points.  python tools/team_sq_md.py gpurun_out/pmc_team_<tag> <tag> <radius>"""
import collections, csv, glob, os, re, sys

src, tag, radius = sys.argv[1], sys.argv[2], sys.argv[3]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("a", "b"):
    for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_shot_" not in k:
                continue
            name = re.search(r"k_shot_\w+(<[^>]*>)?", k).group(0)
            rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Counter_Name"] == "SQ_WAVES":
                rows[name]["_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                rows[name]["_wg"].append(int(r["Workgroup_Size"]))
                rows[name]["_grid"].append(int(r["Grid_Size"]))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    from shot_fpfh_amd import _ffi
    build = _ffi.load().sf_version().decode()
except Exception as e:  # noqa: BLE001
    build = f"unknown ({e})"
out = [f"# K5 at radius {radius} (1M uniform points, lists on both sides of 255): counters per launch ({tag}, {build})", "",
       "`tools/pmc_team.sh` (rocprofv3 --pmc, two passes, dispatches serialised by the profiler); SQ cycle counters are in units of 4 cycles; "
       "a workgroup of `k_shot_team` is one keypoint, the waves a list does not need end at once and are counted in `SQ_WAVES`.", "",
       "| kernel | launches averaged (both passes) | µs under the profiler | workgroups | waves | vector instructions | ... per workgroup | "
       "SIMD cycles issuing vector instructions (SQ_ACTIVE_INST_VALU x 4) / (1024 SIMDs x duration x clock) | LDS bank-conflict cycles / LDS-active cycles |",
       "|---|---|---|---|---|---|---|---|---|"]
for name, c in sorted(rows.items()):
    m = lambda k: sum(c[k]) / len(c[k]) if c.get(k) else float("nan")  # noqa: E731
    us, wg, grid = m("_us"), m("_wg"), m("_grid")
    wgs = grid / wg
    clock = m("GRBM_GUI_ACTIVE") / 8 / us if c.get("GRBM_GUI_ACTIVE") else float("nan")  # MHz: cycles per XCD / us
    busy = m("SQ_ACTIVE_INST_VALU") * 4 / (1024 * us * clock) if clock == clock else float("nan")
    out.append(f"| `{name}` | {len(c['_us'])} | {us:.0f} | {wgs:,.0f} | {m('SQ_WAVES'):,.0f} | {m('SQ_INSTS_VALU'):,.0f} | {m('SQ_INSTS_VALU') / wgs:.0f} | "
               f"{busy:.2f} (clock {clock:.0f} MHz) | {m('SQ_LDS_BANK_CONFLICT') / max(m('SQ_ACTIVE_INST_LDS') * 4, 1):.2f} |")
out += ["", "(The clock is `GRBM_GUI_ACTIVE` / 8 XCDs / duration of the counter pass that carries it; the duration is that of the pass that "
        "carries `SQ_WAVES` first -- both under the profiler, a few per cent slower than the untraced launch.  A main-launch workgroup holds two keypoints; "
        "it is launched for every keypoint and returns at once for those of the other launch.)"]
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", f"{tag}_k5_team.md")
open(path, "w").write("\n".join(out) + "\n")
print("\n".join(out))
