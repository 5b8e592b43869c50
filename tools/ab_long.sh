#!/bin/bash
# Long-list forms, same box: for every abl_libs/lib_*.so, the bench step at radii whose lists exceed 255 points.
# usage on the GPU box: tools/ab_long.sh [radius ...]
cd "$(dirname "$0")/.."
cp shot_fpfh_amd/libshotfpfh.so /tmp/keep.so
for f in abl_libs/lib_*.so; do
  cp $f shot_fpfh_amd/libshotfpfh.so
  echo "== $f"; python tools/bench_radii.py "$@" 2>&1 | grep "^r="
done
cp /tmp/keep.so shot_fpfh_amd/libshotfpfh.so
