#!/bin/bash
# A/B of library builds on one rank of an emulated 8-GPU job (abl_libs/lib_*.so), two rounds
cd "$(dirname "$0")/.."
cp shot_fpfh_amd/libshotfpfh.so /tmp/keep.so
for round in 1 2 3; do
  for f in abl_libs/lib_*.so; do
    cp $f shot_fpfh_amd/libshotfpfh.so
    echo -n "$f "; python bench.py --gpus 8 --emulate-rank 3 --no-match --no-parity --no-strong --sustained-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(round(d['ms_per_step'],4), {n:k[n] for n in k if n[:2] in ('k1','k2')})"
  done
done
cp /tmp/keep.so shot_fpfh_amd/libshotfpfh.so
