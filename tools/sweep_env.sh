#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ... -- runs the descriptor bench once per value and prints the kernel times
var=$1; shift
for x in "$@"; do
  env $var=$x python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-match --no-dropin --no-parity --sustained-seconds 0 --no-density --no-defaults 2>/dev/null > /tmp/sweep_$x.json
  python - "$x" /tmp/sweep_$x.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[2]).read()); k=d["kernels_ms_per_step"]
print(sys.argv[1], round(d["ms_per_step"],3), {n:k[n] for n in ("k2_radius_slots","k1_radix_sort","k1_cell_start","k5_shot","k6_spfh","k7_fpfh")})
PY
done
