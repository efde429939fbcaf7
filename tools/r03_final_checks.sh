#!/bin/bash
# round 3, final checks on the GPU box: the whole GPU suite, the differential fuzzers, the CJK probe
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > $out/r03_tests_final.log 2>&1 || { tail -40 $out/r03_tests_final.log; exit 1; }
tail -1 $out/r03_tests_final.log
: > $out/gpu_fuzz_r03.txt
timeout -k 10 400 python tools/gpu_fuzz_long.py 240 301 2>&1 | grep -v amdgpu.ids >> $out/gpu_fuzz_r03.txt; [ ${PIPESTATUS[0]} -eq 0 ] || { tail -3 $out/gpu_fuzz_r03.txt; exit 1; }
echo "fuzz long done"
timeout -k 10 300 python tools/gpu_fuzz.py --seconds 90 --seed 302 2>&1 | grep -v amdgpu.ids >> $out/gpu_fuzz_r03.txt; [ ${PIPESTATUS[0]} -eq 0 ] || { tail -3 $out/gpu_fuzz_r03.txt; exit 1; }
echo "fuzz (small vocabulary) done"
timeout -k 10 300 python tools/gpu_fuzz.py --seconds 90 --seed 303 --vocab bench 2>&1 | grep -v amdgpu.ids >> $out/gpu_fuzz_r03.txt; [ ${PIPESTATUS[0]} -eq 0 ] || { tail -3 $out/gpu_fuzz_r03.txt; exit 1; }
echo "fuzz (bench vocabulary) done"
timeout -k 10 300 python tools/gpu_fuzz.py --seconds 60 --seed 304 --pattern 1 2>&1 | grep -v amdgpu.ids >> $out/gpu_fuzz_r03.txt; [ ${PIPESTATUS[0]} -eq 0 ] || { tail -3 $out/gpu_fuzz_r03.txt; exit 1; }
echo "fuzz (JSON pattern) done"
cat $out/gpu_fuzz_r03.txt
(echo "== cut decomposition on (default)"; timeout -k 10 500 python tools/cjk_probe.py 20000 2>&1 | grep "^runs of"; echo "== TK_FLAT_CUT=0 (round 2's path: records, hand-back beyond 256 bytes)"; TK_FLAT_CUT=0 timeout -k 10 500 python tools/cjk_probe.py 20000 2>&1 | grep "^runs of") > $out/cjk_probe_r03.txt
cat $out/cjk_probe_r03.txt
