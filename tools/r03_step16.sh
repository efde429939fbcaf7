#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_decode.py -x -q > gpurun_out/r03_dec_tests.log 2>&1 || { tail -30 gpurun_out/r03_dec_tests.log; exit 1; }
tail -1 gpurun_out/r03_dec_tests.log
bash tools/r03_step15.sh
python -c "
import json; d = json.loads(open('gpurun_out/dec3.json').read().strip().splitlines()[-1]); print('c3 decode ms', d['decode']['ms'], 'exact', d['decode']['round_trip_exact'])"
