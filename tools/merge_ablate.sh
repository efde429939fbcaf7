#!/bin/bash
# needs the development build with the ablation hooks (make -C tekken-rs_amd ablate): selected through TK_HIP_LIB, the shipped library stays
export TK_HIP_LIB=${GRAFT_REPO_ROOT:-$(pwd)}/tekken-rs_amd/libtekken_hip_ablate.so
# timing-only ablations of the merge kernels (needs a build with -DTKM_ABLATE): tools/merge_ablate.sh "0 256 512 ..." [bench args]
R=${GRAFT_REPO_ROOT:-$(pwd)}
vals=${1:-"0 256 512 768 1024 2048 3840"}
shift
cd /tmp && export TMPDIR=/tmp
for v in $vals; do
  TK_DEBUG_ABLATE=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl_$v -o a -- python3 $R/bench.py --steps 4 --warmup 1 --cpu-passes 0 --extra-legs none --decode-steps 0 --host-steps 0 "$@" > $R/gpurun_out/abl_$v.log 2>&1 || { tail -5 $R/gpurun_out/abl_$v.log; exit 1; }
  python3 - $R/gpurun_out/abl_$v/a_kernel_stats.csv $v <<'PY'
import csv, sys
rows = {r["Name"].split("(")[0]: float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(sys.argv[1]))}
print("ablate", sys.argv[2], {k: round(v, 3) for k, v in rows.items() if k in ("tk_merge_kernel", "tk_merge_wide_kernel")}, flush=True)
PY
done
