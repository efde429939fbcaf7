#!/bin/bash
# VALU / SALU wave-instructions and duration of tk_flat_kernel per prebuilt library variant (deterministic counts: the metric for
# instruction-count work on the flat kernel): tools/ab_valu.sh "<bench args>" libA.so libB.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
args="$1"; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export TK_HIP_LIB=$R/$v   # (the shipped library is never overwritten: tekken-rs_amd/__init__.py loads what TK_HIP_LIB names)
  tag=$(basename $v .so)
  rm -rf $R/gpurun_out/abv_$tag
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/abv_$tag -o v -- python3 $R/bench.py $args --steps 2 --warmup 1 --cpu-passes 0 --extra-legs none --decode-steps 0 --host-steps 0 --single-docs 0 > $R/gpurun_out/abv_$tag.log 2>&1 || { tail -5 $R/gpurun_out/abv_$tag.log; exit 1; }
  python3 - $R/gpurun_out/abv_$tag $tag <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r["Kernel_Name"].startswith("tk_flat_kernel"):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        acc["ms"].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
print(sys.argv[2], {k: round(sum(v) / len(v) / (1 if k == "ms" else 1e6), 3) for k, v in sorted(acc.items())})
PY
done
