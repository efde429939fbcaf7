#!/bin/bash
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root
cp tekken-rs_amd/libtekken_hip.so gpurun_out/lib_keep.so
for v in tools/probe/lib_emit1.so tools/probe/lib_emit2.so tools/probe/lib_emit1.so tools/probe/lib_emit2.so; do
  cp $v tekken-rs_amd/libtekken_hip.so
  for shape in "--kind ascii" "--kind mixed --doc-len 2048"; do
    timeout -k 10 300 python bench.py $shape --docs 1000000 --steps 3 --warmup 1 --cpu-passes 0 --decode-steps 5 --host-steps 0 --single-docs 0 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '$shape', 'decode ms', d['decode']['ms'], d['decode'].get('kernels_ms'), 'exact', d['decode']['round_trip_exact'])" || exit 1
  done
done
cp gpurun_out/lib_keep.so tekken-rs_amd/libtekken_hip.so
