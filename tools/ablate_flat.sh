#!/bin/bash
# needs the development build with the ablation hooks (make -C tekken-rs_amd ablate): selected through TK_HIP_LIB, the shipped library stays
export TK_HIP_LIB=${GRAFT_REPO_ROOT:-$(pwd)}/tekken-rs_amd/libtekken_hip_ablate.so
# timing-only ablations of tk_flat_kernel (results are garbage, only kernel_ms is meaningful): tools/ablate_flat.sh [bench args]
for ab in ${ABLATE_LIST:-0 16 8 1 3 7 2 4}; do
  TK_DEBUG_ABLATE=$ab timeout -k 10 120 python bench.py --steps 5 --warmup 2 --cpu-passes 0 --extra-legs none --decode-steps 0 --host-steps 0 --single-docs 0 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ablate', $ab, 'kernel_ms', d['roofline']['kernel_ms'], 'pipeline_ms', d['roofline']['pipeline_ms'])"
done
