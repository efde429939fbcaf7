#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/r03_tests.log 2>&1 || { tail -40 $out/r03_tests.log; exit 1; }
tail -1 $out/r03_tests.log
timeout -k 10 300 python bench.py --entry node --gpus 1 --docs 500000 --steps 5 --warmup 2 > $out/r03_node1.json 2> $out/r03_node1.err || { tail -5 $out/r03_node1.err; exit 1; }
cat $out/r03_node1.json
TK_NODE_FORCE_RCCL=1 timeout -k 10 300 python bench.py --entry node --gpus 1 --docs 500000 --steps 5 --warmup 2 > $out/r03_node1_loop.json 2> $out/r03_node1_loop.err || { tail -5 $out/r03_node1_loop.err; exit 1; }
cat $out/r03_node1_loop.json
for fit in same heldout; do
timeout -k 10 300 python bench.py --vocab-fit $fit --steps 100 --decode-steps 0 --host-steps 0 --single-docs 0 --cpu-passes 1 --cpu-sample-docs 100000 --cpu-threads 1 > $out/r03_c2_$fit.json 2> $out/r03_c2_$fit.err || { tail -5 $out/r03_c2_$fit.err; exit 1; }
python - $out/r03_c2_$fit.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1][-14:], 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'exact', d.get('bit_exact_vs_cpu'), d.get('vocab_fit'))
PY
done
