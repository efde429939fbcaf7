#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > $out/r03_tests.log 2>&1 || { tail -40 $out/r03_tests.log; exit 1; }
tail -1 $out/r03_tests.log
Q="--decode-steps 0 --host-steps 0 --single-docs 0 --cpu-passes 1 --cpu-sample-docs 20000"
for cfg in "overlap 256" "serial 256" "overlap 128" "overlap 512"; do
  set -- $cfg
  TK_TAIL=$1 TK_LONG_MIN=$2 timeout -k 10 300 python bench.py --kind zipf --docs 500000 --steps 10 --warmup 2 $Q 2> $out/r03_lm.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TK_TAIL $1 TK_LONG_MIN $2', 'ms_per_step', d['ms_per_step'], 'exact', d.get('bit_exact_vs_cpu'), d.get('handed_back_docs'), 'syncs', d.get('host_syncs'))" || { tail $out/r03_lm.err; exit 1; }
done
bash tools/trace_step.sh zipf --kind zipf --docs 500000 --steps 3 --warmup 1 > /dev/null || exit 1
cat gpurun_out/trace_zipf_timeline.txt | grep -v rocclr | head -60
timeout -k 10 600 python bench.py --kind zipf --docs 4000000 --steps 5 --warmup 1 --cpu-passes 1 --cpu-sample-docs 50000 --decode-steps 0 --host-steps 0 --single-docs 0 2> $out/r03_z4m.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('zipf 4M', 'ms_per_step', d['ms_per_step'], 'value', d['value'], 'exact', d.get('bit_exact_vs_cpu'), d.get('handed_back_docs'), 'syncs', d.get('host_syncs'))"
