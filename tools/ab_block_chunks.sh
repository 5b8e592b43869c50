#!/bin/bash
# block build (one rank of 8, emulated): SF_BLOCK_CHUNKS = workgroups of the two whole-cloud passes
cd "$(dirname "$0")/.."
for c in 1024 4096 1024 4096 8192; do
  echo -n "$c "
  SF_BLOCK_CHUNKS=$c python bench.py --gpus 8 --emulate-rank 3 --no-match --no-parity --no-strong --sustained-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(round(d['ms_per_step'],4), {n:k[n] for n in k if n.startswith('k1')})"
done
