#!/bin/bash
# SQ counter pass (own run, no tracing) for the bench command; output under gpurun_out/pmc_sq_<tag>/
TAG=${1:-x}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_sq_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d "$OUT" -o pmc -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --sustained-seconds 0 --no-density --no-defaults $* > "$OUT/log.txt" 2>&1
tail -1 "$OUT/log.txt" | cut -c1-300
