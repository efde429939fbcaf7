#!/bin/bash
# merge-kernel times on the C2 and C3 shapes (kernel trace), after a change to tk_merge_*: tools/quick_merge.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -o c3 -- python3 $R/bench.py --kind mixed --doc-len 2048 --docs 1000000 --steps 3 --warmup 1 --cpu-passes 1 --cpu-sample-docs 5000 --decode-steps 0 --host-steps 0 > $R/gpurun_out/prof_c3.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -o c2 -- python3 $R/bench.py --steps 5 --warmup 1 --cpu-passes 1 --cpu-sample-docs 20000 --decode-steps 0 --host-steps 0 > $R/gpurun_out/prof_c2.log 2>&1 || exit 1
cd $R
python3 - <<'PY'
import csv, json
for tag in ("c3", "c2"):
    rows = {r["Name"].split("(")[0]: float(r["AverageNs"]) / 1e6 for r in csv.DictReader(open("gpurun_out/prof_%s/%s_kernel_stats.csv" % (tag, tag)))}
    d = json.loads([l for l in open("gpurun_out/prof_%s.log" % tag) if l.startswith("{")][-1])
    print(tag, "exact", d.get("bit_exact_vs_cpu"), "ms_per_step", d["ms_per_step"],
          {k: round(v, 3) for k, v in rows.items() if k in ("tk_flat_kernel", "tk_merge_kernel", "tk_merge_wide_kernel", "tk_flat_assemble_kernel")})
PY
