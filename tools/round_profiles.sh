#!/bin/bash
# Everything the round's profiles/ needs, on the GPU box:  tools/round_profiles.sh r01
tag=${1:-r01}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
timeout -k 10 300 python bench.py > $out/bench_${tag}_c2.json 2> $out/bench_${tag}_c2.err || exit 1
echo "c2 done"
timeout -k 10 300 python bench.py --kind mixed --doc-len 2048 --docs 1000000 --steps 5 --warmup 1 --cpu-passes 1 --cpu-sample-docs 20000 --decode-steps 0 --host-steps 0 > $out/bench_${tag}_c3.json 2> $out/bench_${tag}_c3.err || exit 1
echo "c3 done"
timeout -k 10 300 python bench.py --kind zipf --docs 500000 --steps 5 --warmup 1 --cpu-passes 1 --cpu-sample-docs 50000 --decode-steps 0 --host-steps 0 > $out/bench_${tag}_zipf.json 2> $out/bench_${tag}_zipf.err || exit 1
echo "zipf done"
timeout -k 10 200 python tools/load_time.py > $out/load_time_${tag}.json 2> $out/load_time_${tag}.err || exit 1
echo "load time done"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag} -o ${tag} -- python3 $root/bench.py --steps 5 --warmup 1 --cpu-passes 0 --host-steps 0 > $out/prof_${tag}.log 2>&1 || exit 1
echo "kernel trace done"
cd $root && tools/pmc_flat.sh ${tag}
