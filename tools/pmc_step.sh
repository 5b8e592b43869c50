#!/bin/bash
# SQ counters of every kernel of the bench step (two passes, no tracing) -> profiles/<tag>_step_sq.md: tools/pmc_step.sh <tag>
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_step_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-match --no-dropin --no-parity --no-normals --sustained-seconds 0 --no-density --no-defaults"
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/a" -o pmc -- python3 $ARGS > "$OUT/a.log" 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d "$OUT/b" -o pmc -- python3 $ARGS > "$OUT/b.log" 2>&1
python3 "$REPO/tools/step_sq_md.py" "$OUT" "$TAG"
