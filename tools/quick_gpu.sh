#!/bin/bash
# quick GPU check after a kernel change: parity tests of the flat path, then the C2 bench line (kernel_ms, pipeline_ms)
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/quick_tests.log 2>&1 || { tail -30 gpurun_out/quick_tests.log; exit 1; }
tail -1 gpurun_out/quick_tests.log
timeout -k 10 200 python bench.py --cpu-passes 0 --extra-legs none ${BENCH_ARGS} 2> gpurun_out/quick_bench.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']
print('value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'frac', r['frac'], 'exact', d.get('bit_exact_vs_cpu'), 'decode', (d.get('decode') or {}).get('ms'))"
