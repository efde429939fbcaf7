#!/bin/bash
# same-box A / B of one environment switch on the shipped library: tools/ab_env.sh N "<bench args>" VAR=VALUE   (runs without / with the variable, N round-robin repetitions)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
n="$1"; args="$2"; kv="$3"
for rep in $(seq 1 $n); do
  for mode in off on; do
    if [ $mode = on ]; then export "$kv"; else unset "${kv%%=*}"; fi
    timeout -k 10 300 python bench.py $args --cpu-passes 0 --extra-legs none --decode-steps 0 --host-steps 0 --single-docs 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']; print('$kv', '$mode', 'rep', $rep, 'ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'merge_ms', r.get('merge_kernels_ms'))" || exit 1
  done
done
