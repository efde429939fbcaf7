#!/bin/bash
# usage: ab_env.sh "<bench args>" VAR   (A: unset, B: VAR=1)
args="$1"; var="$2"
for rep in 1 2; do
  for val in "" 1; do
    if [ -n "$val" ]; then export $var=$val; else unset $var; fi
    timeout -k 10 300 python bench.py $args --cpu-passes 0 --decode-steps 0 --host-steps 0 --single-docs 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$val', 'rep', $rep, 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], d.get('bit_exact_vs_cpu'))" || exit 1
  done
done
