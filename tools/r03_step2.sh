#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
bash tools/trace_step.sh zipf --kind zipf --docs 500000 --steps 3 --warmup 1 > /dev/null || exit 1
tail -45 gpurun_out/trace_zipf_timeline.txt
bash tools/ab_bench.sh tools/probe/lib_r02.so tools/probe/lib_cut1.so
