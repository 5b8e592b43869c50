#!/bin/bash
# Per-rank step time of an N-GPU job, emulated on ONE GPU (bench.py --emulate-rank: the exchange call itself is skipped), weak
# (N x 1M points) and strong (the 1M-point cloud cut N ways).  Prints a table; run through gpurun.
cd "$(dirname "$0")/.."
python bench.py --no-cpu-baseline --no-match --no-dropin --no-normals --no-parity --sustained-seconds 0 --no-density --no-defaults 2>/dev/null > /tmp/n1.json
python - <<'P'
import json,subprocess,sys
n1=json.load(open('/tmp/n1.json'))['ms_per_step']
print(f"| N | rank | weak ms/step | weak efficiency | strong ms/step | strong speed-up (of N) |")
print("|---|---|---|---|---|---|")
print(f"| 1 | 0 | {n1:.3f} | 1.00 | {n1:.3f} | 1.00 |")
for n in (2,4,8):
    for r in sorted({0, n//2 - (1 if n>2 else 0) if n>2 else 1}):
        out=subprocess.run([sys.executable,'bench.py','--gpus',str(n),'--emulate-rank',str(r),'--no-match','--no-parity','--sustained-seconds','0'],capture_output=True,text=True).stdout
        d=json.loads(out.strip().splitlines()[-1])
        w=d['ms_per_step']; s=d['strong_scaling']['ms_per_step']
        print(f"| {n} | {r} | {w:.3f} | {n1/w:.2f} | {s:.3f} | {n1/s:.2f} |")
P
