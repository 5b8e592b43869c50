#!/usr/bin/env python3
"""tools/pmc_k5.sh's counters of the fused SHOT kernel -> profiles/<tag>_k5_sq.json, stamped with sf_version() of the library
they were measured on (bench.py quotes the file only for that very build).

Per wave (= per keypoint: one wave each): SQ_INSTS_VALU (vector instructions issued), SQ_ACTIVE_INST_VALU (quad-cycles the
SIMD spent issuing them; x 4 = cycles).  The shader clock during the kernel is GRBM_GUI_ACTIVE cycles over the dispatch's
own duration (the counter is summed over the 8 XCDs)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    out_dir, tag = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "x"
    fs = glob.glob(out_dir + "/a/**/*counter_collection.csv", recursive=True)
    if not fs:
        sys.exit("no counters under " + out_dir)
    per = collections.defaultdict(lambda: collections.defaultdict(float))  # dispatch -> counter -> value
    dur = {}
    for r in csv.DictReader(open(fs[0])):
        if "shot_cached" not in r["Kernel_Name"]:
            continue
        per[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
        dur[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    if not per:
        sys.exit("no k_shot_cached dispatch in the counter file")
    n = len(per)
    avg = collections.defaultdict(float)
    for d in per.values():
        for k, v in d.items():
            avg[k] += v / n
    waves = avg["SQ_WAVES"]
    dur_ns = sum(dur.values()) / n
    clock = None
    if avg.get("GRBM_GUI_ACTIVE") and dur_ns > 0:
        clock = avg["GRBM_GUI_ACTIVE"] / dur_ns * 1e3  # MHz if the counter is one clock domain ...
        if clock > 3000.0:
            clock /= 8.0                                 # ... summed over the 8 XCDs
    from shot_fpfh_amd import _ffi

    rec = {
        "_build": _ffi.load().sf_version().decode(),
        "_source": f"profiles/{tag}_k5_sq.json: rocprofv3 --pmc passes of tools/pmc_k5.sh {tag} (k_shot_cached, {n} dispatches under "
                   "the profiler)",
        "waves_per_launch": waves,
        "SQ_INSTS_VALU_per_wave": round(avg["SQ_INSTS_VALU"] / waves, 1),
        "SQ_ACTIVE_INST_VALU_per_wave": round(avg["SQ_ACTIVE_INST_VALU"] / waves, 1),
        "SQ_WAVE_CYCLES_per_wave": round(avg["SQ_WAVE_CYCLES"] / waves, 1),
        "SQ_INSTS_LDS_per_wave": round(avg["SQ_INSTS_LDS"] / waves, 1),
        "kernel_us_under_profiler": round(dur_ns / 1e3, 1),
        "clock_mhz": round(clock, 0) if clock else 2400.0,
        "clock_source": "GRBM_GUI_ACTIVE / dispatch duration" if clock else "assumed (MI355X peak engine clock)",
    }
    json.dump(rec, open(os.path.join(ROOT, "profiles", f"{tag}_k5_sq.json"), "w"), indent=1)
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
