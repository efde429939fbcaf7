#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec3 -o dec3 -- python3 $R/bench.py --kind mixed --doc-len 2048 --docs 1000000 --steps 2 --warmup 1 --cpu-passes 0 --decode-steps 3 --host-steps 0 --single-docs 0 > $R/gpurun_out/dec3.json 2> $R/gpurun_out/dec3.err || exit 1
cd $R && grep -E "decode|scan" gpurun_out/prof_dec3/dec3_kernel_stats.csv | cut -d, -f1-4
