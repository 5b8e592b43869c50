#!/bin/bash
# Round-end evidence on ONE MI355X box (run through gpurun): everything profiles/ quotes for the library that is in place.
#   tools/final_profiles.sh <tag>      e.g. r03
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
cd "$REPO"
mkdir -p gpurun_out
# 1. kernel trace + HBM counters of the bench command -> profiles/<tag>_summary.md, profiles/<tag>_traffic.json (stamped)
bash tools/profile_gpu.sh ${TAG} --no-match --no-normals > gpurun_out/${TAG}_profile.log 2>&1
python tools/parse_rocprof.py ${TAG} > /dev/null 2>> gpurun_out/${TAG}_profile.log
python tools/trace_gaps.py $(find gpurun_out/prof_${TAG}/trace -name '*kernel_trace.csv' | head -1) auto 20 > profiles/${TAG}_trace_gaps.txt 2>&1
# 2. SQ counters of K5 -> profiles/<tag>_k5_sq.json (stamped)
bash tools/pmc_k5.sh ${TAG} > gpurun_out/${TAG}_k5.log 2>&1
# 2b. the same for K5's team form beside the cached form at a radius with lists on both sides of 255 -> profiles/<tag>_k5_team.md
bash tools/pmc_team.sh ${TAG} 0.04 > gpurun_out/${TAG}_k5_team.log 2>&1
python tools/team_sq_md.py gpurun_out/pmc_team_${TAG} ${TAG} 0.04 >> gpurun_out/${TAG}_k5_team.log 2>&1
# 2c. what the SIMDs do during every kernel of the step -> profiles/<tag>_step_sq.md
bash tools/pmc_step.sh ${TAG} > gpurun_out/${TAG}_step_sq.log 2>&1
# 3. the bench record itself (now quoting 1. and 2.), the emulated ranks of an 8-GPU job, config 4
python bench.py > profiles/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err   # the line the driver parses (ONE run: the first of this build)
cp bench_detail.json profiles/${TAG}_bench_detail.json                            # every other block of that run
for r in 0 3 7; do python bench.py --gpus 8 --emulate-rank $r --no-strong > profiles/${TAG}_emulated_rank${r}_of_8.json 2>/dev/null; cp bench_detail_rank${r}_of_8.json profiles/${TAG}_emulated_rank${r}_of_8_detail.json; done
python tools/run_config4.py > profiles/${TAG}_config4.txt 2>&1
# 4. K8 / K9 at config 4's size under rocprofv3 (kernel trace, fabric traffic, L2, SQ incl. the matrix-core busy counter)
#    -> profiles/<tag>_match_summary.md ; the 8-rank dry run on this one GPU -> profiles/<tag>_dry_run_n8.json (+ detail)
bash tools/profile_match_r6.sh ${TAG} 1000000 > gpurun_out/${TAG}_match_profile.log 2>&1
SF_BENCH_DETAIL_DIR=$REPO/profiles python bench.py --dry-run-n 8 > profiles/${TAG}_dry_run_n8.json 2> gpurun_out/${TAG}_dry_run.err
mv profiles/bench_detail_dry_run_n8.json profiles/${TAG}_dry_run_n8_detail.json 2>/dev/null
# 5. the kernels the step does not contain: compute_normals (k_knn4, k_radius_cov / k_pca_cov) and, at a radius with lists on both
#    sides of 255 points, the second launches (k_shot_team, k_fpfh_mcl, k_spfh tail) -> profiles/<tag>_other_kernels.md
bash tools/profile_others.sh ${TAG} > gpurun_out/${TAG}_others.log 2>&1
python tools/show_bench.py profiles/${TAG}_bench.json profiles/${TAG}_emulated_rank0_of_8.json profiles/${TAG}_emulated_rank3_of_8.json profiles/${TAG}_emulated_rank7_of_8.json
tail -12 profiles/${TAG}_config4.txt
mkdir -p gpurun_out/${TAG}_profiles && cp profiles/${TAG}_* gpurun_out/${TAG}_profiles/ 2>/dev/null
tail -3 gpurun_out/${TAG}_k5.log
