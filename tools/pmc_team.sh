#!/bin/bash
# SQ counters of K5's forms at a radius whose lists exceed 255 points (k_shot_team beside k_shot_cached): tools/pmc_team.sh <tag> [radius]
TAG=${1:-x}; R=${2:-0.04}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_team_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --output-format csv -d "$OUT/a" -o pmc -- python3 $REPO/tools/bench_radii.py $R > "$OUT/a.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/b" -o pmc -- python3 $REPO/tools/bench_radii.py $R > "$OUT/b.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
for sub in ("a", "b"):
    fs = glob.glob(sys.argv[1] + f"/{sub}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(sub, "no counters"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(fs[0])):
        agg[r["Kernel_Name"][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": n[r["Kernel_Name"][:70]] += 1
    for k, v in agg.items():
        if "shot_" in k:
            d = n[k]
            print(sub, k[:50], "dispatches", d, {c: round(x / d, 1) for c, x in v.items()})
PY
