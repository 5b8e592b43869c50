python bench.py --overlap --no-cpu-baseline --no-match --no-dropin --no-normals > gpurun_out/r3d_overlap.json 2>/dev/null
python bench.py --no-cpu-baseline --no-match --no-dropin --no-normals > gpurun_out/r3d_plain.json 2>/dev/null
python tools/show_bench.py gpurun_out/r3d_overlap.json gpurun_out/r3d_plain.json
( time python bench.py --gpus 2 --oversubscribe --steps 5 --warmup 2 > gpurun_out/r3d_os2.json 2> gpurun_out/r3d_os2.err ) 2>&1 | grep real
tail -3 gpurun_out/r3d_os2.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r3d_os2.json"))
print(d["ms_per_step"], d["parity"]["ok"], d.get("strong_scaling"), {k:d["exchange_match"][k] for k in ("ms_per_pass","matches_recovering_true_correspondence","exchange")})
P
