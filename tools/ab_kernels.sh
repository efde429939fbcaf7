#!/bin/bash
# same-box A/B of prebuilt library variants with per-kernel times (rocprofv3 kernel trace): tools/ab_kernels.sh "<bench args>" libA.so libB.so ...
R=${GRAFT_REPO_ROOT:-$(pwd)}
args="$1"; shift
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export TK_HIP_LIB=$R/$v   # (the shipped library is never overwritten: tekken-rs_amd/__init__.py loads what TK_HIP_LIB names)
  tag=$(basename $v .so)
  rm -rf $R/gpurun_out/abk_$tag
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abk_$tag -o k -- python3 $R/bench.py $args --steps 3 --warmup 1 --cpu-passes 0 --extra-legs none --decode-steps 0 --host-steps 0 --single-docs 0 > $R/gpurun_out/abk_$tag.log 2>&1 || { tail -5 $R/gpurun_out/abk_$tag.log; exit 1; }
  python3 - $R/gpurun_out/abk_$tag/k_kernel_stats.csv $tag <<'PY'
import csv, sys
rows = {r["Name"].split("(")[0]: float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open(sys.argv[1]))}
print(sys.argv[2], {k: round(v, 3) for k, v in sorted(rows.items(), key=lambda kv: -kv[1]) if v > 0.02})
PY
done
