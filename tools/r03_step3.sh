#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out
Q="--decode-steps 0 --host-steps 0 --single-docs 0 --cpu-passes 1 --cpu-sample-docs 20000"
for lm in 1024 512 256 128; do
  TK_LONG_MIN=$lm timeout -k 10 300 python bench.py --kind zipf --docs 500000 --steps 10 --warmup 2 $Q 2> $out/r03_lm.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TK_LONG_MIN', $lm, 'ms_per_step', d['ms_per_step'], 'exact', d.get('bit_exact_vs_cpu'), d.get('handed_back_docs'))" || exit 1
done
