( time python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) 2>&1 | tail -8
python bench.py > gpurun_out/r3c_n1.json 2> gpurun_out/r3c_n1.err; tail -2 gpurun_out/r3c_n1.err
python bench.py --points-per-gpu 200000 --steps 3 --warmup 1 --no-match --no-cpu-baseline --no-dropin --no-normals --checksum > gpurun_out/r3c_cs1.json 2>gpurun_out/r3c_cs1.err
for n in 2 4; do python bench.py --gpus $n --oversubscribe --points-per-gpu 200000 --steps 3 --warmup 1 --no-match --checksum > gpurun_out/r3c_cs$n.json 2>gpurun_out/r3c_cs$n.err; tail -1 gpurun_out/r3c_cs$n.err; done
python - <<'P'
import json
for n in (1,2,4):
    try:
        d=json.load(open(f"gpurun_out/r3c_cs{n}.json"))
        print(n, d.get("checksum"), (d.get("strong_scaling") or {}).get("checksum"), d["parity"]["ok"], d["config"]["exchange"][:60])
    except Exception as e: print(n, "ERR", e)
P
python tools/show_bench.py gpurun_out/r3c_n1.json
