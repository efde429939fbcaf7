#!/bin/bash
# decode kernels of the C2 bench under rocprofv3 (per-kernel averages) + the round-trip check:  tools/decode_time.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_dec -o dec -- python3 $R/bench.py --steps 5 --warmup 2 --cpu-passes 0 --extra-legs none --host-steps 0 --single-docs 0 > $R/gpurun_out/dec.json 2> $R/gpurun_out/dec.err || exit 1
cd $R && python -c "
import json; d = json.loads(open('gpurun_out/dec.json').read().strip().splitlines()[-1]); print('step', d['ms_per_step'], 'decode ms', d['decode']['ms'], 'round trip exact', d['decode']['round_trip_exact'])" && grep -E "decode" gpurun_out/prof_dec/dec_kernel_stats.csv | cut -d, -f1-4
