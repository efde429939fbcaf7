#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out
bash tools/quick_merge.sh || exit 1
(echo "== cut decomposition on (default)"; timeout -k 10 500 python tools/cjk_probe.py 50000; echo "== TK_FLAT_CUT=0"; TK_FLAT_CUT=0 timeout -k 10 500 python tools/cjk_probe.py 50000) 2>&1 | grep -v amdgpu.ids > $out/r03_cjk_probe.txt || { tail -5 $out/r03_cjk_probe.txt; exit 1; }
cat $out/r03_cjk_probe.txt
