#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for the bench command.
# Usage: tools/profile_gpu.sh <tag> [bench args...]
# Outputs under gpurun_out/prof_<tag>/ ; summarise with tools/parse_rocprof.py into profiles/.
set -u
TAG=${1:-r01}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (the default step counts: 5 warm-up + 20 timed + 5 bracketed steps and the >= 2 s sustained window -- the summary is taken
# from a run of the length bench.py's `sustained` block reports)
ARGS="$REPO/bench.py --no-cpu-baseline --no-dropin --no-parity --no-density --no-defaults $*"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 $ARGS > "$OUT/trace.log" 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1
timeout 900 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -o pmc -- python3 $ARGS > "$OUT/pmc_l2.log" 2>&1
timeout 900 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_clock" -o pmc -- python3 $ARGS > "$OUT/pmc_clock.log" 2>&1
find "$OUT" -name '*.csv' | head -30
tail -2 "$OUT/trace.log"
