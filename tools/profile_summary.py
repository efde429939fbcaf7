#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of a profiling run (gpurun_out/...) into the committed summaries under
profiles/.  Usage: python tools/profile_summary.py <round-tag>   (e.g. r01)

Expects (produced by the commands listed in profiles/README.md):
  gpurun_out/prof_<tag>/<tag>_kernel_stats.csv                 rocprofv3 --kernel-trace --stats
  gpurun_out/pmc_fetch_<tag>/<tag>_counter_collection.csv      rocprofv3 --pmc FETCH_SIZE   (own pass)
  gpurun_out/pmc_write_<tag>/<tag>_counter_collection.csv      rocprofv3 --pmc WRITE_SIZE   (own pass)
  gpurun_out/pmc_sq_<tag>*/<tag>_counter_collection.csv        SQ / TCC counters (own passes)
  gpurun_out/bench_<tag>_c2.json                               the bench line of the same command
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counters(path, kernel_substr):
    agg = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if kernel_substr in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    out = os.path.join(ROOT, "profiles")
    go = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(go, "prof_%s" % tag, "%s_kernel_stats.csv" % tag), os.path.join(out, "%s_kernel_stats.csv" % tag))
    bench = None
    bpath = os.path.join(go, "bench_%s_c2.json" % tag)
    if os.path.exists(bpath):
        with open(bpath) as f:
            for line in f:
                if line.startswith("{"):
                    bench = json.loads(line)
        shutil.copy(bpath, os.path.join(out, "%s_bench_c2.json" % tag))
    n_docs = bench["config"]["docs_total"] if bench else 1_000_000
    n_ids = bench["config"]["ids_total"] if bench else 98_128_307
    fetch = os.path.join(go, "pmc_fetch_%s" % tag, "%s_counter_collection.csv" % tag)
    write = os.path.join(go, "pmc_write_%s" % tag, "%s_counter_collection.csv" % tag)
    res = {"round": tag,
           "command": "rocprofv3 --pmc FETCH_SIZE (and, in a separate pass, WRITE_SIZE) --kernel-trace --output-format csv "
                      "-- python3 bench.py --steps 3 --warmup 1 --cpu-passes 0"}
    raw = {}
    for kname in ("tk_encode_kernel<0>", "tk_compact_kernel", "tk_scan_apply"):
        raw[kname] = {}
        raw[kname].update({k + "_KB": v for k, v in counters(fetch, kname).items()})
        raw[kname].update({k + "_KB": v for k, v in counters(write, kname).items()})
    res["raw_per_launch"] = raw
    # FETCH_SIZE is calibrated on a kernel of this repo that reads a KNOWN byte count with the same 4-byte-per-lane loads
    known = 4 * n_ids + 4 * n_docs + 8 * (n_docs + 1) * 2
    rep = raw["tk_compact_kernel"]["FETCH_SIZE_KB"] * 1024
    cal = known / rep
    res["fetch_calibration"] = {"kernel": "tk_compact_kernel", "known_read_bytes": known, "FETCH_SIZE_bytes": rep, "factor": cal,
                                "why": "MI355X_MICROARCH.md: FETCH_SIZE is exact only for 16-B/lane streams (where it reads 1/2); "
                                       "other widths must be calibrated on a known byte count in the same access pattern"}
    enc = raw["tk_encode_kernel<0>"]
    rd = enc["FETCH_SIZE_KB"] * 1024 * cal
    wr = enc["WRITE_SIZE_KB"] * 1024
    res["tk_encode_kernel_read_bytes_per_launch"] = rd
    res["tk_encode_kernel_write_bytes_per_launch"] = wr
    res["tk_encode_kernel_bytes_per_launch"] = rd + wr
    if bench:
        res["algorithmic_bytes_per_launch"] = bench["roofline"]["bytes_alg_per_launch"]
        res["traffic_over_algorithmic"] = (rd + wr) / bench["roofline"]["bytes_alg_per_launch"]
    with open(os.path.join(out, "hbm_traffic.json"), "w") as f:
        json.dump(res, f, indent=1)
    sq = {}
    for d in sorted(glob.glob(os.path.join(go, "pmc_sq_%s*" % tag))):
        p = os.path.join(d, "%s_counter_collection.csv" % tag)
        if os.path.exists(p):
            sq.update(counters(p, "tk_encode_kernel<0>"))
    if sq:
        with open(os.path.join(out, "%s_sq_counters.json" % tag), "w") as f:
            json.dump({"kernel": "tk_encode_kernel<0>", "per_launch": sq,
                       "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)"}, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "raw_per_launch"}, indent=1))
    print(json.dumps(sq, indent=1))


if __name__ == "__main__":
    main()
