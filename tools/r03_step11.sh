#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $out/r03_tests.log 2>&1 || { tail -40 $out/r03_tests.log; exit 1; }
tail -1 $out/r03_tests.log
bash tools/quick_merge.sh || exit 1
timeout -k 10 300 python tools/gpu_fuzz_long.py 60 91 > $out/r03_fuzz_long2.txt 2>&1 || { tail -5 $out/r03_fuzz_long2.txt; exit 1; }
tail -1 $out/r03_fuzz_long2.txt
timeout -k 10 300 python tools/gpu_fuzz.py --seconds 60 --seed 35 --vocab bench > $out/r03_fuzz2.txt 2>&1 || { tail -5 $out/r03_fuzz2.txt; exit 1; }
tail -1 $out/r03_fuzz2.txt
