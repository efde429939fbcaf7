#!/bin/bash
# like ab_bench_kind.sh with N round-robin repetitions: tools/ab_bench_reps.sh N "<bench args>" libA.so libB.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
n="$1"; shift
args="$1"; shift
cp tekken-rs_amd/libtekken_hip.so gpurun_out/lib_keep.so
for rep in $(seq 1 $n); do
  for v in "$@"; do
    cp $v tekken-rs_amd/libtekken_hip.so
    timeout -k 10 300 python bench.py $args --cpu-passes 0 --decode-steps 0 --host-steps 0 --single-docs 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'rep', $rep, 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])" || exit 1
  done
done
cp gpurun_out/lib_keep.so tekken-rs_amd/libtekken_hip.so
