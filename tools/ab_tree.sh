#!/bin/bash
# Same-box A/B of two whole TREES (library + Python of each), three rounds: this tree against a copy of another commit's, built
# in place under abl_libs/ (gitignored, travels to the GPU box):   git clone . /tmp/x && (cd /tmp/x && git checkout <commit> && make
# -C shot_fpfh_amd/csrc) && cp -r /tmp/x abl_libs/other_tree   --   then on the box: tools/ab_tree.sh abl_libs/other_tree
OTHER=${1:-abl_libs/other_tree}
FLAGS="--steps 20 --warmup 5 --no-cpu-baseline --no-match --no-dropin --no-parity --sustained-seconds 0 --no-density --no-defaults --no-normals"
for round in 1 2 3; do
  for t in . "$OTHER"; do
    echo -n "$t "; (cd $t && python bench.py $FLAGS 2>/dev/null) | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels_ms_per_step']; print(round(d['ms_per_step'],4), {n:k[n] for n in k if n[:2] in ('k2','k5','k6','k7')})"
  done
done
