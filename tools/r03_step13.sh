#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/r03_tests.log 2>&1 || { tail -40 $out/r03_tests.log; exit 1; }
tail -1 $out/r03_tests.log
bash tools/ab_bench_fit.sh tools/probe/lib_head.so tools/probe/lib_p2hot.so
timeout -k 10 300 python bench.py --kind mixed --doc-len 2048 --docs 1000000 --steps 5 --warmup 2 --cpu-passes 0 --decode-steps 3 --host-steps 0 --single-docs 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3', 'ms_per_step', d['ms_per_step'], 'decode', d.get('decode'))"
