#!/bin/bash
# static instruction statistics of the device kernels in tk_flat.hip (no GPU needed): tools/asm_stats.sh [kernel-substring]
root=$(cd $(dirname $0)/.. && pwd)
mkdir -p $root/gpurun_out/asm
cd $root/tekken-rs_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -S --cuda-device-only -o $root/gpurun_out/asm/flat.s csrc/tk_flat.hip 2>/dev/null
python3 - $root/gpurun_out/asm/flat.s "${1:-tk_flat_kernel}" <<'PY'
import re, sys
lines = open(sys.argv[1]).read().split('\n')
want = sys.argv[2]
names = [m.group(1) for m in (re.match(r'^(_Z\w+):', l) for l in lines) if m]
for nm in names:
    if want not in nm: continue
    start = next(i for i, l in enumerate(lines) if l.startswith(nm + ':'))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith('\t.section') or 'uses_flat_scratch' in lines[i])
    body = [l.strip() for l in lines[start + 1:end]]
    ins = [l for l in body if l and not l.startswith(('.', ';'))]
    v = sum(1 for l in ins if l.startswith('v_'))
    rl = sum(1 for l in ins if l.startswith(('v_readlane', 'v_writelane')))
    print(nm, 'instr', len(ins), 'VALU', v, 'of which read/writelane', rl, 'SALU', sum(1 for l in ins if l.startswith('s_')),
          'scratch', sum(1 for l in ins if l.startswith('scratch_')))
for l in lines:
    if re.search(r'\.(vgpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|name):', l) and 'args' not in l:
        print(l.strip())
PY
