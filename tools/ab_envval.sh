#!/bin/bash
# same-box comparison of one environment variable's values on one bench shape:  tools/ab_envval.sh "<bench args>" VAR v1 v2 ...  ("-" = unset)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
args="$1"; var="$2"; shift; shift
for rep in 1 2; do
  for val in "$@"; do
    if [ "$val" = "-" ]; then unset $var; else export $var=$val; fi
    timeout -k 10 400 python bench.py $args --cpu-passes 0 --extra-legs none --decode-steps 0 --host-steps 0 --single-docs 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$val', 'rep', $rep, 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'handed back', d.get('handed_back_docs'))" || exit 1
  done
done
