#!/bin/bash
# round-3 dev loop on the GPU box: parity tests, the Zipf share with / without the cut decomposition, same-box C2 A/B
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > $out/r03_tests.log 2>&1 || { tail -40 $out/r03_tests.log; exit 1; }
tail -1 $out/r03_tests.log
Q="--decode-steps 0 --host-steps 0 --single-docs 0"
for cut in 1 0; do
  TK_FLAT_CUT=$cut timeout -k 10 300 python bench.py --kind zipf --docs 500000 --steps 10 --warmup 2 --cpu-passes 1 --cpu-sample-docs 50000 $Q > $out/r03_zipf_cut$cut.json 2> $out/r03_zipf_cut$cut.err || { tail -20 $out/r03_zipf_cut$cut.err; exit 1; }
  python - $out/r03_zipf_cut$cut.json <<'PY'
import sys, json
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print('zipf', sys.argv[1][-9:-5], 'value', d['value'], 'ms_per_step', d['ms_per_step'], 'exact', d.get('bit_exact_vs_cpu'), {k: d.get(k) for k in ('handed_back_docs', 'long_piece_records', 'cut_chunks')})
PY
done
bash tools/ab_bench.sh tools/probe/lib_r02.so tools/probe/lib_cut1.so
