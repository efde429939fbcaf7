#!/bin/bash
# Hardware-counter passes over bench.py (C2), one counter group per pass (no tracing domains besides --kernel-trace).
# Usage (on the GPU box): tools/pmc_flat.sh <tag>      -> gpurun_out/pmc_<tag>_<group>/
tag=${1:-flat}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  timeout -k 10 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $root/gpurun_out/pmc_${tag}_${name} -o $tag -- \
    python3 $root/bench.py --steps 3 --warmup 1 --cpu-passes 0 --decode-steps 0 --host-steps 0 --single-docs 0 > $root/gpurun_out/pmc_${tag}_${name}.log 2>&1 || return 1
  echo "pass $name done"
}
run fetch FETCH_SIZE && run write WRITE_SIZE && \
run sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH && \
run sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS && \
run tcc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE
