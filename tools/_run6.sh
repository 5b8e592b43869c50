python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed"
python bench.py --no-cpu-baseline --no-match --no-dropin > gpurun_out/r3e_n1.json 2>/dev/null
for r in 0 3; do python bench.py --gpus 8 --emulate-rank $r --no-strong > gpurun_out/r3e_emul$r.json 2>/dev/null; done
python tools/show_bench.py gpurun_out/r3e_n1.json gpurun_out/r3e_emul0.json gpurun_out/r3e_emul3.json
python tools/bench_voxel_icp.py 2>&1 | tail -5
