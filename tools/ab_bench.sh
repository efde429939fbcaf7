#!/bin/bash
# same-box A/B of prebuilt library variants on the plain bench line (no profiler): [AB_STEPS=40] [AB_ARGS="--vocab-fit heldout"] tools/ab_bench.sh libA.so libB.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2 3; do
  for v in "$@"; do
    export TK_HIP_LIB=$R/$v   # (the shipped library is never overwritten: tekken-rs_amd/__init__.py loads what TK_HIP_LIB names)
    timeout -k 10 200 python bench.py --steps ${AB_STEPS:-40} --warmup 5 $AB_ARGS --cpu-passes 0 --extra-legs none --decode-steps 0 --host-steps 0 --single-docs 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']
print('$v', 'rep', $rep, 'ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'merge_ms', r.get('merge_kernels_ms'), 'exact', d.get('bit_exact_vs_cpu'))" || exit 1
  done
done
