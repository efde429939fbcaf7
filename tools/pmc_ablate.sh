#!/bin/bash
# needs the development build with the ablation hooks (make -C tekken-rs_amd ablate): selected through TK_HIP_LIB, the shipped library stays
export TK_HIP_LIB=${GRAFT_REPO_ROOT:-$(pwd)}/tekken-rs_amd/libtekken_hip_ablate.so
# VALU / SALU / LDS instruction counts of tk_flat_kernel per timing ablation (TK_DEBUG_ABLATE): the difference between
# two ablations is the instruction count of the phase between them.   tools/pmc_ablate.sh  -> gpurun_out/pmc_abl_<n>/
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for ab in ${ABLATE_LIST:-16 8 1 64 128 2 0}; do
  export TK_DEBUG_ABLATE=$ab
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv \
    -d $root/gpurun_out/pmc_abl_$ab -o abl -- python3 $root/bench.py ${BENCH_ARGS} --steps 2 --warmup 1 --cpu-passes 0 --extra-legs none --decode-steps 0 --host-steps 0 --single-docs 0 > $root/gpurun_out/pmc_abl_$ab.log 2>&1 || exit 1
  python3 - $root/gpurun_out/pmc_abl_$ab $ab <<'PY'
import csv, glob, sys, collections
d, ab = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith(("tk_flat_kernel", "tk_flat_dbg_kernel", "tk_flat_split_kernel")):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc["ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print("ablate", ab, {k: round(sum(v) / len(v) / 1e6, 2) for k, v in sorted(acc.items())}, "(millions per launch)")
PY
done
