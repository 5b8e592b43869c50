#!/bin/bash
# A/B of K1 variants: prints the k1_* kernels and the step (tools/ab_libs.sh prints K2..K7)
cd "$(dirname "$0")/.."
cp shot_fpfh_amd/libshotfpfh.so /tmp/keep.so
for round in 1 2; do
  for f in abl_libs/lib_*.so; do
    cp $f shot_fpfh_amd/libshotfpfh.so
    echo -n "$f "; python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-match --no-dropin --no-normals --sustained-seconds 0 --no-density --no-defaults "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(round(d['ms_per_step'],4), d['parity']['ok'], {n:k[n] for n in k if n[:2] in ('k1','k6')})"
  done
done
cp /tmp/keep.so shot_fpfh_amd/libshotfpfh.so
