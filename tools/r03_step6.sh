#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out
timeout -k 10 500 python tools/gpu_fuzz_long.py 240 31 > $out/r03_fuzz_long.txt 2>&1 || { tail -5 $out/r03_fuzz_long.txt; exit 1; }
tail -2 $out/r03_fuzz_long.txt
timeout -k 10 300 python tools/gpu_fuzz.py 120 32 > $out/r03_fuzz.txt 2>&1 || { tail -5 $out/r03_fuzz.txt; exit 1; }
tail -2 $out/r03_fuzz.txt
(echo "== cut decomposition on (default)"; timeout -k 10 300 python tools/cjk_probe.py 100000; echo "== TK_FLAT_CUT=0"; TK_FLAT_CUT=0 timeout -k 10 300 python tools/cjk_probe.py 100000) > $out/r03_cjk_probe.txt 2>&1 || { tail -5 $out/r03_cjk_probe.txt; exit 1; }
cat $out/r03_cjk_probe.txt
