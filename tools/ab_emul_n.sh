#!/bin/bash
# per-kernel times of one emulated rank at N = 1, 2, 4, 8 (weak): where the per-rank step grows with N
cd "$(dirname "$0")/.."
python bench.py --no-cpu-baseline --no-match --no-dropin --no-normals --no-parity --sustained-seconds 0 --no-density --no-defaults 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print(1, 0, round(d['ms_per_step'],4), k)"
for n in 2 4 8; do
  r=$((n/2))
  python bench.py --gpus $n --emulate-rank $r --no-match --no-parity --no-strong --sustained-seconds 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']; print($n, $r, round(d['ms_per_step'],4), k)"
done
