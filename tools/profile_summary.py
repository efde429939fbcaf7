#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/round_profiles.sh (gpurun_out/...) into the committed summaries under
profiles/.  Usage: python tools/profile_summary.py <round-tag>   (e.g. r01)

Expects:
  gpurun_out/prof_<tag>/<tag>_kernel_stats.csv                 rocprofv3 --kernel-trace --stats
  gpurun_out/pmc_<tag>_fetch/<tag>_counter_collection.csv      rocprofv3 --pmc FETCH_SIZE   (own pass)
  gpurun_out/pmc_<tag>_write/<tag>_counter_collection.csv      rocprofv3 --pmc WRITE_SIZE   (own pass)
  gpurun_out/pmc_<tag>_sq1|sq2|tcc/...                         SQ / TCC counters (own passes)
  gpurun_out/bench_<tag>_{c2,c3,zipf}.json                     bench lines
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("tk_flat_kernel", "tk_merge_kernel", "tk_merge_wide_kernel", "tk_flat_assemble_kernel", "tk_flat_counts_kernel", "tk_flat_firstdoc_kernel",
           "tk_scan_block_sums", "tk_scan_top", "tk_scan_apply", "tk_merge_wavefirst_kernel")


def counters(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return agg
    with open(path) as f:
        for r in csv.DictReader(f):
            agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
    out = os.path.join(ROOT, "profiles")
    go = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    shutil.copy(os.path.join(go, "prof_%s" % tag, "%s_kernel_stats.csv" % tag), os.path.join(out, "%s_kernel_stats.csv" % tag))
    bench = None
    for cfg in ("c2", "c2_heldout", "c3", "c3_vocab50k", "zipf", "zipf_serial_tail", "zipf4m"):
        bpath = os.path.join(go, "bench_%s_%s.json" % (tag, cfg))
        if not os.path.exists(bpath):
            continue
        with open(bpath) as f, open(os.path.join(out, "%s_bench_%s.json" % (tag, cfg)), "w") as g:
            for line in f:
                if line.startswith("{"):
                    # (round 3: the PMC passes are taken per shape -- tools/pmc_flat.sh <tag> [<suffix> <shape>] -- and bench.py quotes
                    # the file of ITS shape; a line of any other shape carries null by itself)
                    g.write(line)
                    if cfg == "c2":
                        bench = json.loads(line)
    lt = os.path.join(go, "load_time_%s.json" % tag)
    if os.path.exists(lt):
        shutil.copy(lt, os.path.join(out, "%s_load_time.json" % tag))
    fz = os.path.join(go, "gpu_fuzz_%s.txt" % tag)
    if os.path.exists(fz):
        shutil.copy(fz, os.path.join(out, "%s_gpu_fuzz.txt" % tag))
    lp = os.path.join(go, "longpiece_time_%s.txt" % tag)
    if os.path.exists(lp):
        shutil.copy(lp, os.path.join(out, "%s_longpiece_time.txt" % tag))
    cj = os.path.join(go, "cjk_probe_%s.txt" % tag)
    if os.path.exists(cj):
        with open(cj) as f, open(os.path.join(out, "%s_cjk_probe.txt" % tag), "w") as g:
            g.writelines(l for l in f if l.startswith("runs of") or l.startswith("=="))
    # kernel stats and one-step timelines of the other shapes (tools/trace_step.sh <tag>_c3 / <tag>_zipf)
    for cfg in ("c3", "zipf"):
        for f in glob.glob(os.path.join(go, "trace_%s_%s" % (tag, cfg), "**", "*_kernel_stats.csv"), recursive=True):
            shutil.copy(f, os.path.join(out, "%s_kernel_stats_%s.csv" % (tag, cfg)))
        tl = os.path.join(go, "trace_%s_%s_timeline.txt" % (tag, cfg))
        if os.path.exists(tl):
            with open(tl) as f, open(os.path.join(out, "%s_timeline_%s.txt" % (tag, cfg)), "w") as g:
                g.writelines(l for l in f if "rocclr" not in l or "step span" in l)
    benches = {}
    for cfg in ("c2", "c3", "zipf"):
        bp = os.path.join(out, "%s_bench_%s.json" % (tag, cfg))
        if os.path.exists(bp):
            with open(bp) as f:
                benches[cfg] = json.loads(f.readline())
    # one traffic / counter summary per shape the PMC passes were taken on (tools/pmc_flat.sh <tag> [<suffix> <shape>])
    for sfx, cfg, tname in (("", "c2", "hbm_traffic.json"), ("_c3", "c3", "hbm_traffic_c3.json"), ("_zipf", "zipf", "hbm_traffic_zipf.json")):
        summarize(tag, sfx, benches.get(cfg), os.path.join(out, tname), os.path.join(out, "%s_sq_counters%s.json" % (tag, sfx)), go)


KERNELS_ALL = KERNELS + ("tk_flat_cut_kernel", "tk_flat_todo_kernel", "tk_flat_long_kernel", "tk_flat_long128_kernel", "tk_flat_long_coop_kernel",
                         "tk_encode_kernel<3>", "tk_long_walk_kernel", "tk_long_merge_kernel", "tk_long_compact_kernel")


def summarize(tag, sfx, bench, tpath, sqpath, go):
    dirs = [d for d in sorted(glob.glob(os.path.join(go, "pmc_%s%s_*" % (tag, sfx)))) if os.path.isdir(d)]
    if sfx == "":
        dirs = [d for d in dirs if os.path.basename(d).count("_") == 2]      # pmc_<tag>_<group> only, not the suffixed shapes
    if not dirs:
        return
    n_docs = bench["config"]["docs_total"] if bench else 1_000_000
    n_ids = bench["config"]["ids_total"] if bench else 98_128_307
    allc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for k, cs in counters(os.path.join(d, "%s_counter_collection.csv" % tag)).items():
            k = k.replace("void ", "")
            for cname, v in cs.items():
                allc[k][cname].extend(v)
    mean = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in allc.items()}
    git = (bench or {}).get("build", {}).get("git")
    try:   # the commit of the library the passes ran on (written by __graft_entry__.build(); the same file bench.py stamps its line with)
        with open(os.path.join(ROOT, "tekken-rs_amd", "BUILD_INFO.json")) as f:
            git = json.load(f).get("git") or git
    except Exception:  # noqa: BLE001
        pass
    res = {"round": tag, "git": git, "shape": (bench or {}).get("config", {}).get("workload"),
           "command": "tools/pmc_flat.sh %s %s...: one rocprofv3 --pmc <group> --kernel-trace --output-format csv pass per counter group over "
                      "python3 bench.py --steps 3 --warmup 1 --cpu-passes 0 --decode-steps 0 <shape>" % (tag, sfx),
           "raw_per_launch_KB": {k: {c: mean[k][c] for c in ("FETCH_SIZE", "WRITE_SIZE") if c in mean.get(k, {})} for k in KERNELS_ALL if k in mean}}
    # FETCH_SIZE calibration (MI355X_MICROARCH.md: exact 1/2 for 16-B/lane streams, other widths must be calibrated on a
    # known byte count): the assembly kernel reads a known number of bytes with 4-B/lane loads
    asm = mean.get("tk_flat_assemble_kernel", {})
    if "FETCH_SIZE" in asm:
        known = 4 * (n_ids - 2 * n_docs) + 16 * n_docs + 8 * n_docs   # id slots (holes ~2 % more) + doc records + output offsets
        rep = asm["FETCH_SIZE"] * 1024
        res["fetch_calibration"] = {"kernel": "tk_flat_assemble_kernel", "known_read_bytes_lower_bound": known,
                                    "FETCH_SIZE_bytes": rep, "factor_4B_per_lane": known / rep, "factor_16B_per_lane": 2.0}
    fk = mean.get("tk_flat_kernel", {})
    if "FETCH_SIZE" in fk and "WRITE_SIZE" in fk:
        # the flat kernel streams the text with 16-B/lane loads (factor 2) and probes tables with 16-B random loads
        # (uncalibrated, factor 2 taken as the upper bound)
        rd = fk["FETCH_SIZE"] * 1024 * 2.0
        wr = fk["WRITE_SIZE"] * 1024
        res["tk_flat_kernel_read_bytes_per_launch_upper"] = rd
        res["tk_flat_kernel_write_bytes_per_launch"] = wr
        res["tk_flat_kernel_bytes_per_launch"] = rd + wr
        if bench:
            res["algorithmic_bytes_per_launch"] = bench["roofline"]["bytes_alg_per_launch"]
            res["traffic_over_algorithmic"] = (rd + wr) / bench["roofline"]["bytes_alg_per_launch"]
    # the whole pipeline of one step: every tokenization kernel, reads at FETCH_SIZE x 2 (the upper bound: exact for the
    # 16-B/lane streams, too high for scattered probes and 4-B/lane loads), writes as counted
    per_kernel = {}
    for k in KERNELS_ALL:
        m = mean.get(k, {})
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            launches_per_step = {"tk_scan_block_sums": 2, "tk_scan_top": 2, "tk_scan_apply": 2}.get(k, 1)
            per_kernel[k] = {"read_upper": m["FETCH_SIZE"] * 1024 * 2.0 * launches_per_step, "write": m["WRITE_SIZE"] * 1024 * launches_per_step}
    if per_kernel:
        res["pipeline_per_kernel_bytes_per_step"] = per_kernel
        res["pipeline_bytes_per_step"] = sum(v["read_upper"] + v["write"] for v in per_kernel.values())
        if bench:
            res["pipeline_traffic_over_algorithmic"] = res["pipeline_bytes_per_step"] / bench["roofline"]["bytes_alg_per_launch"]
    # the VALU-issue floor of the dominant kernel: dynamic VALU wave-instructions (SQ_INSTS_VALU) x the issue cost of the kernel's
    # instruction mix (tools/valu_mix.py: static mix x the per-class cycles tools/ubench/valu2.hip measured at 4..7 waves per SIMD --
    # 2 for 32-bit encoded two-operand ALU instructions, the guide's SIMD-32 figure, ~4.1-4.8 for compares, DPP, three-operand and
    # 64-bit encoded ones) over 1 024 SIMDs at the shader clock the micro-benchmark ran at.  The guide's 2 cycles for EVERY
    # instruction is the lower bound beside it.
    if "SQ_INSTS_VALU" in fk:
        res["tk_flat_kernel_valu_insts_per_launch"] = fk["SQ_INSTS_VALU"]
        clk_mix, ghz, src = 4.0, 2.4, "assumed 4 cycles at 2.4 GHz (profiles/r04_valu_mix.json missing)"
        try:
            with open(os.path.join(ROOT, "profiles", "r04_valu_mix.json")) as f:
                vm = json.load(f)["issue_cost"]
            clk_mix, ghz, src = vm["clk_mix"], vm["shader_GHz"], "profiles/r04_valu_mix.json (tools/valu_mix.py + profiles/ubench/r04_valu2.json)"
        except Exception:  # noqa: BLE001
            pass
        res["tk_flat_kernel_valu_clk_mix"] = clk_mix
        res["tk_flat_kernel_valu_clk_source"] = src
        res["tk_flat_kernel_valu_floor_ms"] = fk["SQ_INSTS_VALU"] * clk_mix / 1024.0 / (ghz * 1e9) * 1e3
        res["tk_flat_kernel_valu_floor_ms_at_2clk"] = fk["SQ_INSTS_VALU"] * 2.0 / 1024.0 / (ghz * 1e9) * 1e3
    with open(tpath, "w") as f:
        json.dump(res, f, indent=1)
    sq = {k: {c: v for c, v in mean.get(k, {}).items() if c not in ("FETCH_SIZE", "WRITE_SIZE")} for k in KERNELS[:4]}
    with open(sqpath, "w") as f:
        json.dump({"git": git, "shape": res["shape"], "per_launch": sq, "note": "SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (MI355X_MICROARCH.md)"}, f, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "raw_per_launch_KB"}, indent=1))


if __name__ == "__main__":
    main()
