#!/bin/bash
# K8 at BASELINE config 4's size (two 1M-point clouds' SHOT rows, tools/run_config4.py) under rocprofv3 on the GPU box:
# kernel-trace + stats, then separate counter passes (no tracing beside them): FETCH_SIZE, WRITE_SIZE, L2 hits / misses, and two
# SQ passes (matrix-core busy cycles, vector issue, LDS, clock).  Summarised by tools/match_r6_md.py into profiles/<tag>_match_summary.md.
# Usage (via gpurun): tools/profile_match_r6.sh <tag> [n_points]
set -u
TAG=${1:-r06}
N=${2:-1000000}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${TAG}_match
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/tools/run_config4.py $N 2000"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 $ARGS > "$OUT/trace.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1
timeout 600 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -o pmc -- python3 $ARGS > "$OUT/pmc_l2.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq_a" -o pmc -- python3 $ARGS > "$OUT/sq_a.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d "$OUT/sq_b" -o pmc -- python3 $ARGS > "$OUT/sq_b.log" 2>&1
grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" "$OUT/trace.log" | tail -12
python3 "$REPO/tools/match_r6_md.py" "$TAG" "$N"
