#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
out=$root/gpurun_out
ABLATE_LIST="0 16 8 1 64 128 3 2" bash tools/ablate_flat.sh --kind mixed --doc-len 2048 --docs 400000 2>&1 | tee $out/r03_ablate_c3.txt
