#!/usr/bin/env python3
"""One line per bench.py JSON file: value, ms/step, roofline fraction and kernel, the per-kernel breakdown, parity."""
import json
import sys

for path in sys.argv[1:]:
    d = json.load(open(path))
    r = d.get("roofline") or {}
    print(f"{path}: {(d.get('value') or d.get('projected_value_upper_bound') or 0) / 1e6:.1f} M/s  {d.get('ms_per_step', 0):.3f} ms/step  n_gpus {d.get('n_gpus')}  "
          f"roofline {r.get('kernel')} {r.get('frac', 0):.3f} ({r.get('avg_launch_ms', 0):.3f} ms)  parity "
          f"{(d.get('parity') or {}).get('ok')}")
    print("   ", d.get("kernels_ms_per_step"))
    extra = {k: (d[k].get("desc_per_s_both") if k == "dropin_host_to_host" else d[k].get("value") if k == "cpu_baseline"
                 else d[k].get("ms_per_pass")) for k in ("dropin_host_to_host", "cpu_baseline", "exchange_match") if d.get(k)}
    if extra:
        print("   ", extra)
