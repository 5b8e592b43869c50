#!/bin/bash
# SQ / LDS counters for the SHOT chain (K4, K5): two passes (counter groups), no tracing
TAG=${1:-x}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_k5_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --only shot --no-match --no-dropin --no-parity --no-normals --sustained-seconds 0 --no-density --no-defaults $*"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d "$OUT/a" -o pmc -- python3 $ARGS > "$OUT/a.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES --output-format csv -d "$OUT/b" -o pmc -- python3 $ARGS > "$OUT/b.log" 2>&1
# which queue the waves of K5 sit in (round 5): waiting cycles by cause, vector-memory writes, the texture cache's stall on pending requests
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d "$OUT/c" -o pmc -- python3 $ARGS > "$OUT/c.log" 2>&1
# (at most two counters of the texture cache per pass: more "exceeds the capabilities of the hardware to collect", and the profiler
# then aborts and hangs in its own signal handler -- every pass under `timeout`)
timeout 300 rocprofv3 --pmc SQ_WAVES TCP_PENDING_STALL_CYCLES_sum --output-format csv -d "$OUT/d" -o pmc -- python3 $ARGS > "$OUT/d.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d "$OUT/e" -o pmc -- python3 $ARGS > "$OUT/e.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum --output-format csv -d "$OUT/f" -o pmc -- python3 $ARGS > "$OUT/f.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
for sub in ("a", "b", "c", "d", "e", "f"):
    fs = glob.glob(sys.argv[1] + f"/{sub}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print(sub, "no counters"); continue
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        if "shot_cached" in k:
            w = v.get("SQ_WAVES", 1.0)
            print(sub, {c: round(x / w, 1) for c, x in v.items()}, "waves", w)
PY
# the numbers bench.py's roofline_all.k5_shot.valu_issue quotes, stamped with the library's build id
python3 "$REPO/tools/k5_sq_json.py" "$OUT" "$TAG"
python3 "$REPO/tools/k5_wait_md.py" "$OUT" "$TAG"
