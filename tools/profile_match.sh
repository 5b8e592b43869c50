#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for the K8 matching bench.
# Usage: tools/profile_match.sh <tag> M1 M2 D ; summarise with
#   python tools/parse_rocprof.py <tag> "python3 tools/bench_match.py M1 M2 D"
set -u
TAG=${1:-r01_match}; shift || true
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="$REPO/tools/bench_match.py $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 $ARGS > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -o pmc -- python3 $ARGS > "$OUT/pmc_l2.log" 2>&1
tail -3 "$OUT/trace.log"
