#!/bin/bash
# SQ / LDS counters for the FPFH chain (K6, K7): two passes (counter groups), no tracing
TAG=${1:-x}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_k7_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --only fpfh --no-match --no-dropin --no-parity --no-normals --sustained-seconds 0 --no-density --no-defaults $*"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT/a" -o pmc -- python3 $ARGS > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU --output-format csv -d "$OUT/b" -o pmc -- python3 $ARGS > "$OUT/b.log" 2>&1
tail -2 "$OUT/b.log" | cut -c1-200
