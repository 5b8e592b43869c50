#!/bin/bash
# A/B timing of library builds on ONE box (boxes differ by ~1 %): abl_libs/lib_*.so are swapped in turn, three rounds.
# usage on the GPU box: tools/ab_libs.sh [bench flags]
cd "$(dirname "$0")/.."
cp shot_fpfh_amd/libshotfpfh.so /tmp/keep.so
for round in 1 2 3; do
  for f in abl_libs/lib_*.so; do
    cp $f shot_fpfh_amd/libshotfpfh.so
    echo -n "$f "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-match --no-dropin --no-parity --sustained-seconds 0 --no-density --no-defaults "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(round(d['ms_per_step'],4), {n:k[n] for n in k if n[:2] in ('k2','k5','k6','k7','k4')})"
  done
done
cp /tmp/keep.so shot_fpfh_amd/libshotfpfh.so
