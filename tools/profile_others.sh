#!/bin/bash
# Kernels outside the headline step, under rocprofv3 (kernel trace + FETCH / WRITE + SQ passes), summarised by tools/others_md.py:
#   compute_normals on the C3 cloud, both branches (tools/bench_normals.py: k_knn4 + k_pca_cov, k_radius + k_pca_cov);
#   the descriptor step at radius 0.04 (tools/bench_radii.py 0.04: lists on both sides of 255 points -> k_shot_team, k_fpfh_mcl, the tails).
set -u
TAG=${1:-r06}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_${TAG}_others
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for what in normals radii; do
  if [ $what = normals ]; then ARGS="$REPO/tools/bench_normals.py"; else ARGS="$REPO/tools/bench_radii.py 0.04"; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${what}_trace" -o trace -- python3 $ARGS > "$OUT/${what}_trace.log" 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/${what}_fetch" -o pmc -- python3 $ARGS > "$OUT/${what}_fetch.log" 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/${what}_write" -o pmc -- python3 $ARGS > "$OUT/${what}_write.log" 2>&1
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/${what}_sq" -o pmc -- python3 $ARGS > "$OUT/${what}_sq.log" 2>&1
done
python3 "$REPO/tools/others_md.py" "$TAG"
