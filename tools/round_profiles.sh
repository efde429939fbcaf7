#!/bin/bash
# Everything the round's profiles/ needs, on the GPU box:  tools/round_profiles.sh r03
# (bench lines for every BASELINE shape, kernel traces + stats, one-step timelines, PMC passes per shape;
# tools/profile_summary.py <tag> turns gpurun_out/ into the committed files under profiles/)
tag=${1:-r04}
part=${2:-all}          # all | profiles (step 1: traces + PMC passes + their summary) | bench (step 2: the bench lines): two calls fit gpurun's 20 minutes each
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out
cd $root
if [ "$part" != bench ]; then
# 1. kernel traces and PMC passes first; their summary (profiles/hbm_traffic*.json, stamped with this build's commit) is written on
#    the box so that the bench lines of step 2 carry traffic and VALU floor measured at the SAME commit
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_${tag} -o ${tag} -- python3 $root/bench.py --steps 20 --warmup 2 --cpu-passes 0 --extra-legs none --host-steps 0 --single-docs 0 > $out/prof_${tag}.log 2>&1 || exit 1
echo "kernel trace (c2) done"
cd $root
bash tools/trace_step.sh ${tag}_c3 --kind mixed --doc-len 2048 --docs 1000000 --steps 3 --warmup 1 > /dev/null || exit 1
echo "kernel trace + timeline (c3) done"
bash tools/trace_step.sh ${tag}_zipf --kind zipf --docs 500000 --steps 3 --warmup 1 > /dev/null || exit 1
echo "kernel trace + timeline (zipf) done"
tools/pmc_flat.sh ${tag} "" || exit 1
tools/pmc_flat.sh ${tag} _c3 --kind mixed --doc-len 2048 --docs 1000000 || exit 1
tools/pmc_flat.sh ${tag} _zipf --kind zipf --docs 500000 || exit 1
for cfg in c2 c3 zipf; do   # (the summary wants each shape's algorithmic byte count: a short bench line)
  case $cfg in c2) A="";; c3) A="--kind mixed --doc-len 2048";; zipf) A="--kind zipf --docs 500000";; esac
  [ -f profiles/${tag}_bench_$cfg.json ] || timeout -k 10 300 python bench.py $A --steps 3 --warmup 1 --cpu-passes 0 --extra-legs none --decode-steps 0 --host-steps 0 --single-docs 0 > profiles/${tag}_bench_$cfg.json 2>/dev/null
done
python tools/profile_summary.py ${tag} > $out/profile_summary_${tag}.log 2>&1 || { tail -5 $out/profile_summary_${tag}.log; exit 1; }
mkdir -p $out/profiles_${tag} && cp profiles/hbm_traffic*.json profiles/${tag}_sq_counters*.json $out/profiles_${tag}/
echo "PMC summary written"
fi
[ "$part" = profiles ] && exit 0
# 2. the bench lines
Q="--decode-steps 0 --host-steps 0 --single-docs 0"
timeout -k 10 300 python bench.py > $out/bench_${tag}_c2.json 2> $out/bench_${tag}_c2.err || exit 1
echo "c2 done"
timeout -k 10 300 python bench.py --vocab-fit heldout --steps 100 --cpu-passes 1 --cpu-sample-docs 100000 --cpu-threads 1 $Q > $out/bench_${tag}_c2_heldout.json 2> $out/bench_${tag}_c2_heldout.err || exit 1
echo "c2 (held-out vocabulary) done"
timeout -k 10 300 python bench.py --vocab-fit heldout --fresh-batches 25 --steps 20 --warmup 5 --cpu-passes 1 --cpu-sample-docs 100000 --cpu-threads 1 $Q > $out/bench_${tag}_c2_heldout_memo.json 2> $out/bench_${tag}_c2_heldout_memo.err || exit 1
echo "c2 (held-out vocabulary, memo of merged pieces on, 25 fresh batches) done"
timeout -k 10 300 python bench.py --fresh-batches 25 --steps 20 --warmup 5 --cpu-passes 0 --extra-legs none $Q > $out/bench_${tag}_c2_fresh.json 2> $out/bench_${tag}_c2_fresh.err || exit 1
echo "c2 (fitted vocabulary, fresh batches: the adaptive policy leaves the memo alone) done"
timeout -k 10 300 python bench.py --kind mixed --doc-len 2048 --docs 1000000 --steps 10 --warmup 2 --cpu-passes 1 --cpu-sample-docs 20000 --decode-steps 3 --host-steps 0 --single-docs 0 > $out/bench_${tag}_c3.json 2> $out/bench_${tag}_c3.err || exit 1
echo "c3 done"
timeout -k 10 300 python bench.py --kind zipf --docs 500000 --steps 10 --warmup 2 --cpu-passes 1 --cpu-sample-docs 50000 $Q > $out/bench_${tag}_zipf.json 2> $out/bench_${tag}_zipf.err || exit 1
echo "zipf (500 k documents = one GPU's share of BASELINE configs[4]) done"
TK_TAIL=serial timeout -k 10 300 python bench.py --kind zipf --docs 500000 --steps 10 --warmup 2 --cpu-passes 0 --extra-legs none $Q > $out/bench_${tag}_zipf_serial_tail.json 2> $out/bench_${tag}_zipf_serial_tail.err || exit 1
echo "zipf with the tail behind the merge kernels (TK_TAIL=serial: the A / B of the second stream) done"
timeout -k 10 600 python bench.py --kind zipf --docs 4000000 --steps 5 --warmup 1 --cpu-passes 1 --cpu-sample-docs 50000 $Q > $out/bench_${tag}_zipf4m.json 2> $out/bench_${tag}_zipf4m.err || exit 1
echo "zipf (4 M documents = all of configs[4] on one GPU) done"
timeout -k 10 200 python tools/load_time.py > $out/load_time_${tag}.json 2> $out/load_time_${tag}.err || exit 1
echo "load time done"
timeout -k 10 200 python tools/gpu_longpiece_time.py > $out/longpiece_time_${tag}.txt 2> /dev/null || exit 1
echo "long piece timing done"
timeout -k 10 300 python tools/pattern_time.py > $out/pattern_time_${tag}.json 2> /dev/null || exit 1
KIND=mixed DOC_LEN=2048 N_DOCS=100000 timeout -k 10 300 python tools/pattern_time.py >> $out/pattern_time_${tag}.json 2> /dev/null || exit 1
echo "JSON pattern timing done"
echo "all done"
