#!/bin/bash
# Hardware-counter passes over bench.py, one counter group per pass (no tracing domains besides --kernel-trace).
# Usage (on the GPU box): tools/pmc_flat.sh <tag> [<suffix> <bench.py arguments of the shape ...>]   -> gpurun_out/pmc_<tag><suffix>_<group>/
#   tools/pmc_flat.sh r03                                             the C2 shape (bench.py's default)
#   tools/pmc_flat.sh r03 _c3 --kind mixed --doc-len 2048 --docs 1000000
#   tools/pmc_flat.sh r03 _zipf --kind zipf --docs 500000
tag=${1:-flat}
sfx=${2:-}
shift; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $root/gpurun_out/pmc_${tag}${sfx}_${name} -o $tag -- \
    python3 $root/bench.py --steps 3 --warmup 1 --cpu-passes 0 --extra-legs none --decode-steps 0 --host-steps 0 --single-docs 0 "${SHAPE[@]}" > $root/gpurun_out/pmc_${tag}${sfx}_${name}.log 2>&1 || return 1
  echo "pass $name$sfx done"
}
SHAPE=("$@")
run fetch FETCH_SIZE && run write WRITE_SIZE && \
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH && \
run sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS && \
run tcc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
