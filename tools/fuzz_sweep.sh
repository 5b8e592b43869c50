#!/bin/bash
# Long differential sweep (GPU vs oracle) over many seeds of tests/test_hip_fuzz.py; run on the GPU box:
#   tools/fuzz_sweep.sh 3:60      -> seeds 3 .. 59
cd "$(dirname "$0")/.."
SF_FUZZ_SEEDS=${1:-3:40} python -m pytest tests/test_hip_fuzz.py -m gpu -q --no-header -x -p no:cacheprovider 2>&1 | grep -v "^RCCL\|^ROCm\|^Host\|^Librccl\|^HIP" | tail -15
