set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for kb in 2048 4096 8192; do
  export SF_MATCH_HALF_CHUNK_KB=$kb
  OUT=$REPO/gpurun_out/prof_r03_match_$kb
  mkdir -p $OUT
  ARGS="$REPO/tools/bench_match.py 262144 262144 352"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 $ARGS > "$OUT/trace.log" 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o pmc -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o pmc -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -o pmc -- python3 $ARGS > "$OUT/pmc_l2.log" 2>&1
  grep -E "^K8" $OUT/trace.log
done
cd $REPO
for kb in 2048 4096 8192; do python tools/parse_rocprof.py r03_match_$kb "SF_MATCH_HALF_CHUNK_KB=$kb python3 tools/bench_match.py 262144 262144 352" > /dev/null 2>&1; grep -E "k8_match_half|k8_half_final" profiles/r03_match_${kb}_summary.md | head -5; done
cp profiles/r03_match_*_summary.md gpurun_out/
