#!/bin/bash
# timeline of ONE step (every kernel launch with its start offset and duration) from a rocprofv3 kernel trace:
#   tools/trace_step.sh <tag> <bench.py arguments ...>      -> gpurun_out/trace_<tag>/..., gpurun_out/trace_<tag>_timeline.txt
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/trace_$tag -o $tag -- python3 $R/bench.py "$@" --cpu-passes 0 --extra-legs none --decode-steps 0 --host-steps 0 --single-docs 0 > $R/gpurun_out/trace_$tag.log 2>&1 || { tail -20 $R/gpurun_out/trace_$tag.log; exit 1; }
cd $R
python3 - $tag <<'PY' | tee gpurun_out/trace_$tag\_timeline.txt
import csv, sys, glob
tag = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob("gpurun_out/trace_%s/**/%s_kernel_trace.csv" % (tag, tag), recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last step: from the last tk_flat_firstdoc_kernel on
names = [r["Kernel_Name"].split("(")[0] for r in rows]
starts = [i for i, n in enumerate(names) if n == "tk_flat_firstdoc_kernel"]
i0 = starts[-1]
t0 = int(rows[i0]["Start_Timestamp"])
end = 0
for r, n in zip(rows[i0:], names[i0:]):
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    end = max(end, e)
    print("%9.1f us  +%8.1f us  %s  grid %s" % (s / 1e3, (e - s) / 1e3, n, r.get("Grid_Size", "")))
print("step span %.1f us" % (end / 1e3))
PY
