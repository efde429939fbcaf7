#!/usr/bin/env python3
"""bench.py -- input MB/s tokenized on MI355X for the tekken-rs `Tekkenizer::encode` hot path.

  python bench.py --gpus N --steps K --warmup W
  N > 1 either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...
  bench.py --gpus N ...: RANK / WORLD_SIZE come from the environment) or bare: `python bench.py --gpus N` starts its own N
  ranks as child processes before anything touches the GPU and relays rank 0's line.

A "step" is one pass of the hot path (split + merge + emit, ids packed in document order) over one
batch of synthetic documents already resident in HBM:
  N = 1 : BASELINE.json configs[1] = 1 M x 512-byte ASCII documents (G-ascii)
  N > 1 : configs[3] shape = 1 M x 512-byte documents PER GPU (weak scaling: 8 M over 8 GPUs),
          contiguous shards, plus the single RCCL gather of the id buffers to rank 0 inside the step.
One JSON line is printed by rank 0 (contract in the task statement) with two extra objects:
  roofline     -- the dominant kernel (tk_flat_kernel: split + lookup + merge, one wave per 2048-byte region)
                  against the HBM roofline:
                  algorithmic bytes (N_in + 8(D+1) + 4 T_out + 8(D+1)) per launch divided by
                  its mean duration measured live with HIP events on the launch stream
  cpu_baseline -- the CPU oracle (a restatement, "port": the reference is Rust and cannot be built
                  here) timed single-thread on this box's host cores on the same documents; cpu_baseline_nt: the same on
                  all host cores (count stated)
Other objects: decode (row f-1), host_to_host (row f-4, PCIe inclusive), single_doc (the reference's one-&str-per-call shape,
BASELINE configs[0]), node (gather-inclusive and kernels-only MB/s, per-GPU kernel time), build (git commit of the library).
--kind mixed = configs[2], --kind zipf = configs[4] (N > 1: documents sharded by BYTES).
"""
import argparse
import collections
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec)

# BASELINE.json configs[] index and shape per --kind (the label the judge keys on)
CONFIG_OF_KIND = {"ascii": ("configs[1]", "configs[3]"), "mixed": ("configs[2]", "configs[2] shape per GPU"),
                  "zipf": ("configs[4] shape on one GPU", "configs[4]")}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start one rank per GPU as ordinary child processes.  Runs BEFORE anything in
    this process touches the GPU (no torch import, no HIP call): the children are plain subprocesses, nothing is re-executed.
    The children are watched together: when one of them dies the others are stopped at once (a rank that waits in a collective
    for a dead peer would otherwise sit there until the collective's own time-out), and the whole launch has a time limit."""
    import socket
    import subprocess
    import time
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        # rank 0 prints the line; the other ranks keep their stderr (a failing rank must be able to say why) but not their stdout
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)   # (rank 0 never blocks on a full pipe)
    reader.start()
    deadline = time.time() + float(os.environ.get("TK_BENCH_LAUNCH_TIMEOUT", "3000"))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = "rank %d exited with code %d" % (r, p.returncode)
        if failed is None and time.time() > deadline:
            failed = "the ranks did not finish in time"
        if failed is None:
            time.sleep(0.2)
    if failed is None:
        for r, p in enumerate(procs):
            if p.returncode != 0:
                failed = "rank %d exited with code %d" % (r, p.returncode)
    if failed is not None:
        for p in procs:                      # exactly the processes started above, by handle
            if p.poll() is None:
                p.terminate()
        t_kill = time.time() + 10
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_kill - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
        raise SystemExit("bench.py --gpus %d: %s; the other ranks were stopped" % (args.gpus, failed))
    reader.join(timeout=30)
    sys.stdout.write(b"".join(chunks).decode())
    sys.stdout.flush()
    raise SystemExit(0)


def build_info():
    """git commit the library was built at (written by __graft_entry__.build(); the GPU box has no .git)."""
    try:
        with open(os.path.join(ROOT, "tekken-rs_amd", "BUILD_INFO.json")) as f:
            return json.load(f)
    except Exception:  # noqa: BLE001
        return {"git": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (0 = default: 400 for ascii = 0.5 s of kernels, 30 otherwise)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--docs", type=int, default=1_000_000, help="documents per GPU")
    ap.add_argument("--doc-len", type=int, default=512)
    ap.add_argument("--kind", default="ascii", choices=["ascii", "mixed", "zipf"])
    ap.add_argument("--cpu-passes", type=int, default=3, help="oracle passes over the CPU sample (0 = skip)")
    ap.add_argument("--cpu-sample-docs", type=int, default=1_000_000)
    ap.add_argument("--vocab", default=os.environ.get("TEKKEN_JSON", ""))
    ap.add_argument("--entry", choices=["ranks", "node"], default="ranks",
                    help="ranks: one process per GPU (torch.distributed, what the driver launches); node: ONE process, the C ABI's "
                         "tk_node_* over GPUs 0..N-1 with pinned host buffers -- what a Rust host calls (PCIe inclusive: never `value` of the contract)")
    ap.add_argument("--vocab-fit", choices=["same", "heldout"], default="same",
                    help="synthetic vocabulary: trained on the corpus's whole word list (same) or with ~15 %% of its word occurrences withheld (heldout)")
    ap.add_argument("--gather", choices=["overlap", "sync"], default="overlap",
                    help="N > 1: gather of batch k beside the kernels of batch k + 1 (default) or inside the step, blocking")
    ap.add_argument("--wire", type=int, choices=[18, 32], default=18, help="N > 1: bits per id on the wire")
    ap.add_argument("--host-steps", type=int, default=3, help="host-to-host leg (row f-4): timed passes, 0 = skip")
    ap.add_argument("--decode-steps", type=int, default=5, help="extra leg: GPU batch decode of the produced ids (0 = skip)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo = dry run of the N>1 code path with several ranks sharing one GPU (ids staged through host)")
    ap.add_argument("--single-docs", type=int, default=1000, help="single_doc leg: sequential one-document calls (C1 shape: 64-byte strings), 0 = skip")
    ap.add_argument("--fresh-batches", type=int, default=0,
                    help="B >= warmup + steps distinct seeded batches resident in HBM, step k runs on batch k, and the memo of merged pieces "
                         "(tk_ctx_set_memo) is ON: the only protocol under which a persistent cache may be timed.  0 (default): one batch repeated, memo OFF")
    ap.add_argument("--extra-legs", default="auto", help="auto: at N = 1 on the default shape also run compact legs for configs[2], one GPU's share of "
                                                           "configs[4] and the held-out vocabulary (fresh batches, memo on and off); none: skip")
    ap.add_argument("--cpu-threads", type=int, default=0, help="cpu_baseline_nt: oracle threads (0 = all host cores, capped at 256; 1 = skip)")
    args = ap.parse_args()
    if args.steps <= 0:
        args.steps = 400 if args.kind == "ascii" else 30
    if args.entry == "node":
        return node_entry(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and os.environ.get("TK_BENCH_FORCE_DIST") != "1":
        self_launch(args)

    import torch
    import torch.distributed as dist
    import corpus
    import synth_vocab as sv
    tk = importlib.import_module("tekken-rs_amd")
    par = importlib.import_module("tekken-rs_amd.parallel")

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    distributed = world > 1 or os.environ.get("TK_BENCH_FORCE_DIST") == "1"   # (the override: the N > 1 code path on one GPU)
    if args.gpus != world and distributed:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    n_dev = max(1, torch.cuda.device_count())
    if distributed and world > n_dev and args.dist_backend == "nccl":
        # fewer GPUs than ranks (a rehearsal on a one-GPU box): RCCL refuses two ranks on one device, gloo carries the gather
        args.dist_backend = "gloo"
    local_rank = local_rank % n_dev
    torch.cuda.set_device(local_rank)
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    xdev = "cuda" if args.dist_backend == "nccl" else "cpu"  # where the collectives' tensors live

    # ---- vocabulary (real tekken.json if TEKKEN_JSON is set, else the seeded synthetic one) ----
    if args.vocab:
        vocab_path, vocab_kind = args.vocab, "tekken.json"
    elif args.vocab_fit == "heldout":
        # a vocabulary that never saw ~15 % of the corpus's word occurrences (tools/synth_vocab.py heldout_words): the miss rate
        # of a real vocabulary on real text, not the 2.4 % of one trained on the very word list the corpus draws from
        vocab_path, vocab_kind = sv.ensure_heldout(), "synthetic-130072-heldout"
    else:
        vocab_path, vocab_kind = sv.ensure_default(), "synthetic-130072"
    tokz = tk.Tekkenizer.from_file(vocab_path, device=local_rank)  # loader + table build + upload
    eng = tokz.engine()

    # ---- this rank's shard of the corpus, generated in place (documents are seeded per index) ----
    seed = corpus.BASE_SEED + (1 if not distributed else 3)
    first_doc, n_docs = rank * args.docs, args.docs
    sharding = "contiguous whole documents per GPU"
    if distributed and args.kind == "zipf":
        # C5: documents of 16 B .. 32 KiB -- contiguous ranges cut so that BYTES are balanced (parallel.shard_by_bytes); every
        # rank derives the same cuts from the lengths alone and generates only its own documents
        all_offs = corpus.offsets(args.kind, args.docs * world, args.doc_len, seed=seed)
        cuts = par.shard_by_bytes(all_offs, world)
        first_doc, n_docs = cuts[rank], cuts[rank + 1] - cuts[rank]
        sharding = "contiguous whole documents per GPU, cut by bytes (parallel.shard_by_bytes)"
        del all_offs
    data, offs = corpus.generate(args.kind, n_docs, args.doc_len, seed=seed, first_doc=first_doc)
    n_bytes = int(offs[-1])
    d_bytes = torch.from_numpy(data).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    stream = torch.cuda.current_stream().cuda_stream
    torch.cuda.synchronize()
    # The memo of merged pieces is a persistent cache: on ONE repeated batch it would answer every unknown piece from the second step
    # on -- work skipped in the timed region.  So: repeated batch (the default) => memo OFF; --fresh-batches B => B distinct batches
    # (seeds apart), step k on batch k, memo on with its adaptive policy, and the line says so.
    fresh = []
    if args.fresh_batches:
        if args.fresh_batches < args.warmup + args.steps:
            raise SystemExit("--fresh-batches %d < warmup + steps = %d: a batch would repeat" % (args.fresh_batches, args.warmup + args.steps))
        if distributed and args.kind == "zipf":
            raise SystemExit("--fresh-batches is not wired for the byte-sharded zipf shape")
        for k in range(args.fresh_batches):
            dk, ok = corpus.generate(args.kind, n_docs, args.doc_len, seed=seed + 1000 * (k + 1), first_doc=first_doc, threads=min(64, os.cpu_count() or 1))
            fresh.append((torch.from_numpy(dk).cuda(), torch.from_numpy(ok.astype(np.int64)).cuda(), int(ok[-1])))
        data, offs, n_bytes = dk, ok, int(ok[-1])           # (the last batch: what the CPU legs and the bit-exact check look at)
        d_bytes, d_offs = fresh[-1][0], fresh[-1][1]
        torch.cuda.synchronize()
    eng.set_memo(int(os.environ.get("TK_MEMO_LOG2", "24")) if fresh else 0, 1 if os.environ.get("TK_MEMO_POLICY") == "always" else 0)
    step_no = [0]
    bytes_seen = [0]

    enc_ms = []
    pipe_ms = []
    merge_ms = []
    n_ids_local = 0
    gathered = None
    # N > 1: the link into rank 0 bounds the job.  The ids travel in the 18-bit wire format when the
    # vocabulary allows it, and the gather of batch k runs beside the kernels of batch k + 1 (at most two in flight);
    # every gather is complete before the clock stops.
    codec = None
    if distributed and xdev == "cuda" and args.wire == 18 and tokz.vocab_size() <= (1 << 18):
        codec = par.Ids18Codec(tk, eng)
    overlap = distributed and args.gather == "overlap"
    pending = collections.deque()

    def drain(keep):
        nonlocal gathered
        while len(pending) > keep:
            gathered = pending.popleft().result()

    if distributed and (overlap or codec is not None):
        # a small rehearsal of the packed / overlapped gather; if it raises on ANY rank, every rank falls back to the
        # plain form (32-bit ids, blocking) -- the ranks agree through an all_reduce so that nobody is left waiting
        ok = 1
        try:
            t_ids = (torch.arange(1000, dtype=torch.int32, device="cuda") * 131 % 200000).to(xdev)
            t_cnt = torch.full((10,), 100, dtype=torch.int64, device=xdev)
            res = par.gather_ids(t_ids, t_cnt, dst=0, codec=codec, wait=not overlap)
            res = res.result() if overlap else res
            if rank == 0:
                exp = torch.cat([t_ids] * world)
                if not (res[0].numel() == exp.numel() and bool(torch.equal(res[0], exp))):
                    ok = 0
        except Exception as e:  # noqa: BLE001
            sys.stderr.write("[bench] packed / overlapped gather failed in the rehearsal (%r): falling back\n" % (e,))
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=xdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            codec, overlap = None, False

    def step():
        nonlocal n_ids_local, gathered
        if fresh:
            fb, fo, fn = fresh[step_no[0] % len(fresh)]
            step_no[0] += 1
            v_ids, v_oo = eng.encode_batch_device_views(fb.data_ptr(), fo.data_ptr(), n_docs, fn, True, True, stream)
            bytes_seen[0] += fn
        else:
            v_ids, v_oo = eng.encode_batch_device_views(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, n_bytes, True, True, stream)
            bytes_seen[0] += n_bytes
        n_ids_local = v_ids.__cuda_array_interface__["shape"][0]
        t = eng.last_timing()
        enc_ms.append(t["encode_kernel_ms"])
        pipe_ms.append(t["pipeline_ms"])
        merge_ms.append(t.get("merge_ms", 0.0))
        if distributed:
            ids = torch.as_tensor(v_ids, device="cuda")
            oo = torch.as_tensor(v_oo, device="cuda")
            cnt = oo[1:] - oo[:-1]
            if overlap:
                # (a private copy: the context's output buffer belongs to the next call)
                pending.append(par.gather_ids(ids.clone().to(xdev), cnt.to(xdev), dst=0, codec=codec, wait=False))
                drain(1)
            else:
                gathered = par.gather_ids(ids.to(xdev), cnt.to(xdev), dst=0, codec=codec)
        return v_ids, v_oo

    for _ in range(args.warmup):
        step()
    drain(0)
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    enc_ms.clear()
    pipe_ms.clear()
    merge_ms.clear()
    bytes_seen[0] = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        v_ids, v_oo = step()
    drain(0)
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tot = torch.tensor([bytes_seen[0] // args.steps, n_ids_local, n_docs], dtype=torch.int64, device=xdev)
        dist.all_reduce(tot)
        total_bytes, total_ids, total_docs = int(tot[0].item()), int(tot[1].item()), int(tot[2].item())
        # per-GPU device time of the tokenization kernels (HIP events around the pipeline): load balance, and the node
        # rate with the gather taken out
        mine = torch.tensor([float(np.mean(pipe_ms)), float(n_bytes)], dtype=torch.float64, device=xdev)
        per_rank = [torch.zeros(2, dtype=torch.float64, device=xdev) for _ in range(world)]
        dist.all_gather(per_rank, mine)
        per_rank = [p.tolist() for p in per_rank]
    else:
        total_bytes, total_ids, total_docs = bytes_seen[0] // args.steps, n_ids_local, n_docs
        per_rank = [[float(np.mean(pipe_ms)), float(n_bytes)]]

    if rank == 0 and distributed:
        g_ids, g_offs = gathered
        assert g_ids.numel() == total_ids and g_offs.numel() == total_docs + 1 and int(g_offs[-1]) == total_ids
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = total_bytes / 1e6 / (elapsed / args.steps)
        # roofline of the dominant kernel on THIS rank: algorithmic bytes per launch / mean launch duration
        bytes_alg = n_bytes + 8 * (n_docs + 1) + 4 * n_ids_local + 8 * (n_docs + 1)
        k_ms = float(np.mean(enc_ms))
        achieved = bytes_alg / (k_ms * 1e-3) / 1e9
        traffic, traffic_info = None, {}
        # (the PMC passes were taken on three shapes -- C2, C3 and one GPU's share of the Zipf shape, with the default vocabulary --:
        # any other shape, size or vocabulary reports null)
        tname = None
        if not distributed and not args.vocab and args.vocab_fit == "same":
            if args.kind == "ascii" and n_docs == 1_000_000 and args.doc_len == 512:
                tname = "hbm_traffic.json"
            elif args.kind == "mixed" and n_docs == 1_000_000 and args.doc_len == 2048:
                tname = "hbm_traffic_c3.json"
            elif args.kind == "zipf" and n_docs == 500_000:
                tname = "hbm_traffic_zipf.json"
        tpath = os.path.join(ROOT, "profiles", tname or "none")
        bound_by = None
        if tname and os.path.exists(tpath):
            try:
                with open(tpath) as f:
                    tj = json.load(f)
                traffic = tj.get("tk_flat_kernel_bytes_per_launch")
                traffic_info = {"traffic_pipeline": tj.get("pipeline_bytes_per_step"), "traffic_measured_at": tj.get("git"),
                                "traffic_source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, this shape)" % tname}
                if tj.get("tk_flat_kernel_valu_floor_ms"):
                    # floor = SQ_INSTS_VALU (PMC pass at the stamped commit) x the measured issue cost of the kernel's instruction mix
                    # (tools/valu_mix.py, profiles/ubench/r04_valu2.json) / (1 024 SIMDs x the measured shader clock); frac = floor / measured.
                    # floor_ms_at_2clk: every instruction at the guide's 2 cycles -- the kernel cannot be faster than that either
                    bound_by = {"unit": "valu", "insts_per_launch": tj["tk_flat_kernel_valu_insts_per_launch"],
                                "clk_per_instruction_mix": tj.get("tk_flat_kernel_valu_clk_mix", 4.0),
                                "floor_ms": round(tj["tk_flat_kernel_valu_floor_ms"], 4),
                                "frac": round(tj["tk_flat_kernel_valu_floor_ms"] / k_ms, 4),
                                "floor_ms_at_2clk": round(tj.get("tk_flat_kernel_valu_floor_ms_at_2clk", 0.0), 4),
                                "measured_at": tj.get("git")}
            except Exception:  # noqa: BLE001
                traffic = None
        cfg1, cfgN = CONFIG_OF_KIND[args.kind]
        shape = {"ascii": "%d x %d-byte ASCII docs (G-ascii)" % (n_docs, args.doc_len),
                 "mixed": "%d x %d-byte mixed UTF-8 docs (G-mixed)" % (n_docs, args.doc_len),
                 "zipf": "%d Zipf-length docs, 16 B - 32 KiB (G-zipf)" % n_docs}[args.kind]
        protocol = ("%d fresh seeded batches resident in HBM, step k on batch k, memo of merged pieces ON (adaptive)" % len(fresh)) if fresh else \
            "one batch repeated, memo of merged pieces OFF"
        if not distributed:
            workload = "%s, 1 x MI355X = BASELINE %s; %s" % (shape, cfg1, protocol)
        else:
            workload = "%s per GPU = BASELINE %s, gather of the id buffers to rank 0 in the step (%s backend, %s, %d-bit ids on the wire); %s" % (
                shape, cfgN, "RCCL" if args.dist_backend == "nccl" else "gloo: ranks share a GPU", "overlap" if overlap else "sync", 18 if codec else 32, protocol)
        k_times = [p[0] for p in per_rank]
        kernel_only = total_bytes / 1e6 / (max(k_times) * 1e-3) if max(k_times) > 0 else None
        out = {
            "metric": "input MB/s tokenized (whole node)", "value": round(value, 1), "unit": "MB/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "baseline_config": cfg1 if not distributed else cfgN,
                       "docs_total": total_docs, "input_bytes_total": total_bytes, "ids_total": total_ids,
                       "vocab": vocab_kind, "add_bos": True, "add_eos": True, "sharding": sharding},
            # achieved = algorithmic bytes of the batch / the time of ALL its kernels, first to last (HIP events around the
            # pipeline on its stream) -- `frac` is that; the dominant kernel alone (the same bytes over ITS duration: it neither writes the
            # final ids nor the output offsets, so this flatters) is kept as kernel_frac
            "roofline": dict({"bound": "hbm", "achieved": round(bytes_alg / (float(np.mean(pipe_ms)) * 1e-3) / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(bytes_alg / (float(np.mean(pipe_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic_info.pop("traffic_pipeline", None),
                              "span": "every kernel of a step, first to last (pipeline_ms)", "pipeline_ms": round(float(np.mean(pipe_ms)), 4),
                              "bytes_alg_per_launch": bytes_alg,
                              "kernel": "tk_flat_kernel", "kernel_ms": round(k_ms, 4), "kernel_achieved": round(achieved, 2),
                              "kernel_frac": round(achieved / HBM_PEAK_GBS, 5), "kernel_traffic": traffic,
                              # (the span from the end of that kernel to the end of both merge kernels, scans included: on text with many pieces
                              # outside the vocabulary -- configs[2] -- the merge kernels are the longest part of a step)
                              "merge_kernels_ms": round(float(np.mean(merge_ms)), 4) if merge_ms else None,
                              "longest_part": "tk_flat_kernel" if not merge_ms or k_ms >= float(np.mean(merge_ms)) else "tk_merge_kernel + tk_merge_wide_kernel",
                              "pipeline_frac": round(bytes_alg / (float(np.mean(pipe_ms)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                              "bound_by": bound_by}, **traffic_info),
            "tokens_per_s": round(total_ids / (elapsed / args.steps), 1),
            "handed_back_docs": eng.last_stats()["handed_back"], "long_piece_records": eng.long_piece_records(), "cut_chunks": eng.cut_chunks(), "host_syncs": eng.last_host_syncs(),
            "node": {"MBps_gather_inclusive": round(value, 1), "MBps_kernels_only": None if kernel_only is None else round(kernel_only, 1),
                     "per_gpu_kernel_ms": [round(t, 4) for t in k_times], "per_gpu_kernel_ms_max_over_mean": round(max(k_times) / (sum(k_times) / len(k_times)), 4) if sum(k_times) > 0 else None,
                     "per_gpu_input_bytes": [int(p[1]) for p in per_rank]},
            "build": build_info(),
        }
        if not distributed:
            # the ids of the last timed step, on the host (the device views die with the next call on the context)
            h_ids = torch.as_tensor(v_ids, device="cuda").cpu().numpy().view(np.uint32)
            h_oo = torch.as_tensor(v_oo, device="cuda").cpu().numpy().astype(np.uint64)
        if not distributed and args.decode_steps > 0:
            out["decode"] = decode_leg(args, tk, eng, v_ids, v_oo, n_docs, n_bytes, d_bytes, stream)
        if not distributed and args.host_steps > 0:
            out["host_to_host"] = host_leg(args, tk, eng, data, offs, h_ids, h_oo)
        if not distributed and args.single_docs > 0:
            out["single_doc"] = single_doc_leg(args, tk, eng, vocab_path)
        out["memo"] = dict(eng.memo_stats(), protocol=protocol)
        if not distributed and args.cpu_passes > 0:
            out.update(cpu_baseline(args, data, offs, vocab_path, h_ids, h_oo, n_bytes))
        if distributed and args.cpu_passes > 0:
            # rank 0's shard on rank 0's host cores, bounded (the contract: N = 1 only needs it; the line keeps it for every N)
            v_i = torch.as_tensor(v_ids, device="cuda").cpu().numpy().view(np.uint32)
            v_o = torch.as_tensor(v_oo, device="cuda").cpu().numpy().astype(np.uint64)
            a2 = argparse.Namespace(**vars(args))
            a2.cpu_sample_docs, a2.cpu_passes, a2.cpu_threads = min(args.cpu_sample_docs, 200_000), 1, 1
            out.update(cpu_baseline(a2, data, offs, vocab_path, v_i, v_o, n_bytes))
        if distributed:
            out["link_model"] = link_model(world, per_rank, total_ids, total_docs, 18 if codec else 32, ms_per_step)
        if not distributed and args.extra_legs == "auto" and args.kind == "ascii" and not args.vocab and args.vocab_fit == "same" and not fresh:
            out["two_in_flight"] = in_flight_leg(tk, vocab_path, eng, d_bytes, d_offs, n_docs, n_bytes, ms_per_step, calls=100 if n_docs >= 500_000 else 10)
        if not distributed and args.extra_legs == "auto" and args.kind == "ascii" and not args.vocab and args.vocab_fit == "same" \
                and n_docs == 1_000_000 and args.doc_len == 512 and not fresh:
            tokz.close()
            tokz = None
            del d_bytes, d_offs
            torch.cuda.empty_cache()
            out.update(extra_legs(args, tk))
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if tokz is not None:
        tokz.close()


def link_model(world, per_rank, total_ids, total_docs, wire_bits, ms_per_step):
    """What DESIGN.md section 5 predicts for this line, so that a measured N > 1 step can be read against it: every peer sends its ids
    (wire_bits each) and per-document counts (u32) straight to rank 0 over its own xGMI link, all peers at once; rank 0's kernels run
    beside the transfers (overlap) -- a step costs max(kernels, slowest link)."""
    ids_per_peer = total_ids / max(1, world)
    docs_per_peer = total_docs / max(1, world)
    bytes_per_peer = ids_per_peer * wire_bits / 8 + docs_per_peer * 4
    link_gbs = 64.0                       # one xGMI link, one direction (MI355X_MICROARCH.md: ~153 GB/s both ways, 60-75 one way)
    link_ms = bytes_per_peer / (link_gbs * 1e9) * 1e3 if world > 1 else 0.0
    k_ms = max(p[0] for p in per_rank)
    return {"peers": world - 1, "bytes_per_peer": int(bytes_per_peer), "assumed_link_GBps_one_way": link_gbs, "expected_link_ms": round(link_ms, 3),
            "kernels_ms_max": round(k_ms, 3), "expected_step_ms_overlap": round(max(k_ms, link_ms), 3),
            "expected_step_ms_sync": round(k_ms + link_ms, 3), "measured_step_ms": round(ms_per_step, 3),
            "note": "the gather the north star prescribes bounds N > 1, not the kernels (DESIGN.md section 5)"}


def shape_leg(tk, vocab_path, kind, n_docs, doc_len, steps, warmup=1, fresh=0, memo_log2=0, memo_policy=0, sample_docs=2000, seed_off=1):
    """One compact line for another single-GPU configuration: `steps` timed passes (HIP events around the pipeline) over a batch
    resident in HBM -- or, with fresh > 0, over `fresh` distinct seeded batches, one per pass, with the memo of merged pieces on
    (cold = the first pass, warm = the passes from the third on) --, a sample of documents id for id against the oracle."""
    import torch
    import corpus
    import synth_vocab as sv
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tk_oracle
    tokz = tk.Tekkenizer.from_file(vocab_path, device=0)
    eng = tokz.engine()
    eng.set_memo(memo_log2, memo_policy)
    stream = torch.cuda.current_stream().cuda_stream
    toks, ns, bos, eos = sv.load_tokens(vocab_path)
    orc = tk_oracle.Oracle(toks, ns, bos, eos)
    rows, exact, alg = [], True, 0
    n_b = max(1, fresh)
    for b in range(n_b):
        data, offs = corpus.generate(kind, n_docs, doc_len, seed=corpus.BASE_SEED + seed_off + 1000 * b, threads=min(64, os.cpu_count() or 1))
        n_bytes = int(offs[-1])
        d_b = torch.from_numpy(data).cuda()
        d_o = torch.from_numpy(offs.astype(np.int64)).cuda()
        reps = 1 if fresh else warmup + steps
        for r in range(reps):
            v_ids, v_oo = eng.encode_batch_device_views(d_b.data_ptr(), d_o.data_ptr(), n_docs, n_bytes, True, True, stream)
            t = eng.last_timing()
            m = eng.memo_stats()
            rows.append({"batch": b, "pipeline_ms": t["pipeline_ms"], "flat_ms": t["encode_kernel_ms"], "merge_ms": t["merge_ms"],
                         "lookups": m["lookups_last"], "hits": m["hits_last"], "active": m["active_last"], "timed": bool(fresh) or r >= warmup,
                         "n_bytes": n_bytes})
        n_ids = v_ids.__cuda_array_interface__["shape"][0]
        alg = n_bytes + 16 * (n_docs + 1) + 4 * n_ids
        if b == n_b - 1 or b == 0:
            k = min(sample_docs, n_docs)
            h_oo = torch.as_tensor(v_oo, device="cuda")[:k + 1].cpu().numpy().astype(np.uint64)
            h_ids = torch.as_tensor(v_ids, device="cuda")[:int(h_oo[k])].cpu().numpy().view(np.uint32)
            eids, eoo = orc.encode_batch(data[:int(offs[k])], offs[:k + 1], True, True, threads=8)
            exact = exact and bool(np.array_equal(h_oo, eoo) and np.array_equal(h_ids, eids))
            st = orc.miss_stats(data[:int(offs[k])], offs[:k + 1])
        del d_b, d_o
    stats = eng.last_stats()
    tokz.close()
    timed = [r for r in rows if r["timed"]]
    warm = [r for r in timed if r["batch"] >= 2] if fresh else timed
    ms = float(np.mean([r["pipeline_ms"] for r in warm]))
    nb = float(np.mean([r["n_bytes"] for r in warm]))
    out = {"docs": n_docs, "input_bytes": int(nb), "steps_timed": len(warm), "ms": round(ms, 4), "MBps": round(nb / 1e6 / (ms * 1e-3), 1),
           "pipeline_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5), "flat_kernel_ms": round(float(np.mean([r["flat_ms"] for r in warm])), 4),
           "merge_kernels_ms": round(float(np.mean([r["merge_ms"] for r in warm])), 4), "handed_back_docs": stats["handed_back"],
           "miss_rate": round(st["missed"] / max(1, st["pieces"]), 4), "bit_exact_vs_cpu_sample_docs": min(sample_docs, n_docs), "bit_exact_vs_cpu": exact}
    if fresh:
        lk, ht = sum(r["lookups"] for r in warm), sum(r["hits"] for r in warm)
        out.update({"protocol": "%d fresh seeded batches, one pass each, memo ON (policy %s); warm = batches 2..%d" % (fresh, "always" if memo_policy else "adaptive", fresh - 1),
                    "cold_first_batch_ms": round(rows[0]["pipeline_ms"], 4), "second_batch_ms": round(rows[1]["pipeline_ms"], 4) if len(rows) > 1 else None,
                    "memo_hit_rate_warm": round(ht / max(1, lk), 4), "memo_active_warm": all(r["active"] for r in warm)})
    else:
        out["protocol"] = "one batch repeated, memo OFF"
    return out


def in_flight_leg(tk, vocab_path, eng, d_bytes, d_offs, n_docs, n_bytes, serial_ms, calls=100):
    """What a host that keeps TWO batches in flight gets out of one GPU: a second context (contexts are independent: own tables, scratch,
    streams), two host threads, each `calls` whole calls on the same resident batch (memo off).  The small kernels and the two host
    waits of one call run beside the big kernels of the other.  Not `value`: that stays one call at a time."""
    import threading
    import torch
    tok2 = tk.Tekkenizer.from_file(vocab_path, device=0)
    engs = [eng, tok2.engine()]
    engs[1].set_memo(0, 0)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    views = [None, None]

    def worker(i, k):
        for _ in range(k):
            views[i] = engs[i].encode_batch_device_views(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, n_bytes, True, True, streams[i].cuda_stream)

    def timed(k):
        ths = [threading.Thread(target=worker, args=(i, k)) for i in range(2)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3 / (2 * k)

    timed(3)
    ms = timed(calls)
    same = all(bool(torch.equal(torch.as_tensor(views[0][j], device="cuda"), torch.as_tensor(views[1][j], device="cuda"))) for j in (0, 1))
    tok2.close()
    return {"ms_per_batch": round(ms, 4), "MBps": round(n_bytes / 1e6 / (ms * 1e-3), 1), "calls_timed": 2 * calls, "one_at_a_time_ms": round(serial_ms, 4),
            "speedup": round(serial_ms / ms, 3), "both_contexts_same_ids": same,
            "protocol": "two contexts, two host threads, the same resident batch, memo OFF; wall clock over all calls / calls"}


def extra_legs(args, tk):
    """The other single-GPU configurations in the driver-run line (compact: a few passes each): BASELINE configs[2], one GPU's share of
    configs[4], and configs[1] with the held-out vocabulary -- memo off on a repeated batch, and memo on over fresh batches."""
    import synth_vocab as sv
    out = {}
    dflt, held = sv.ensure_default(), sv.ensure_heldout()
    out["configs[2]"] = dict(shape_leg(tk, dflt, "mixed", 1_000_000, 2048, steps=3, seed_off=1), workload="1 M x 2 KiB mixed UTF-8 docs (G-mixed), 1 x MI355X = BASELINE configs[2]")
    out["configs[4]_share"] = dict(shape_leg(tk, dflt, "zipf", 500_000, 0, steps=5, seed_off=1), workload="500 k Zipf-length docs 16 B - 32 KiB = one GPU's share of BASELINE configs[4]")
    out["heldout"] = dict(shape_leg(tk, held, "ascii", 1_000_000, 512, steps=10, seed_off=1),
                          workload="configs[1] shape, vocabulary that never saw ~15 % of the word occurrences (--vocab-fit heldout)")
    out["heldout_memo"] = dict(shape_leg(tk, held, "ascii", 1_000_000, 512, steps=0, fresh=8, memo_log2=int(os.environ.get("TK_MEMO_LOG2", "24")), seed_off=1),
                               workload="the same shape and vocabulary, memo of merged pieces on, measured on FRESH batches only")
    return out


def node_entry(args):
    """`--entry node`: the native multi-GPU entry a Rust / C host calls (tk_node_create over GPUs 0..N-1, one process, one host
    thread per device, documents sharded whole by bytes, ONE RCCL gather of the id buffers to GPU 0; csrc/tk_node.cpp), with
    pinned caller buffers (tk_node_encode_batch_pinned).  Host buffers in, host buffers out: the figure includes PCIe both ways
    and the gather -- it is reported under its own metric name, beside the slowest device's kernel time and the exchange time
    from tk_node_last_timing.  Weak scaling: --docs documents per GPU."""
    import importlib
    import time
    import numpy as np
    import corpus
    import synth_vocab as sv
    import tk_oracle
    tk = importlib.import_module("tekken-rs_amd")
    vocab_path = args.vocab or (sv.ensure_heldout() if args.vocab_fit == "heldout" else sv.ensure_default())
    toks, ns, bos, eos = sv.load_tokens(vocab_path)
    n = args.gpus
    n_docs = args.docs * n
    data, offs = corpus.generate(args.kind, n_docs, args.doc_len, seed=corpus.BASE_SEED + 1)
    h_data, h_offs = tk.host_empty(len(data), np.uint8), tk.host_empty(len(offs), np.uint64)
    h_data[:] = data
    h_offs[:] = offs
    h_ids, h_oo = tk.host_empty(len(data) + 2 * len(offs), np.uint32), tk.host_empty(len(offs), np.uint64)
    node = tk.Node(toks, ns, bos, eos, devices=tuple(range(n)))
    for _ in range(max(args.warmup, 1)):
        n_ids = node.encode_batch_into(h_data, h_offs, h_ids, h_oo, True, True)
    k_ms, g_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        n_ids = node.encode_batch_into(h_data, h_offs, h_ids, h_oo, True, True)
        t = node.last_timing()
        k_ms.append(t["kernels_ms_max"])
        g_ms.append(t["gather_ms"])
    elapsed = time.perf_counter() - t0
    # parity on a sample through the oracle (the checker), ids in document order as from one GPU
    sample = min(n_docs, 20000)
    orc = tk_oracle.Oracle(toks, ns, bos, eos)
    eids, eoo = orc.encode_batch(data[:int(offs[sample])], offs[:sample + 1], True, True, threads=16)
    exact = bool(np.array_equal(h_ids[:int(h_oo[sample])], eids) and np.array_equal(h_oo[:sample + 1], eoo))
    ms = elapsed / args.steps * 1e3
    print(json.dumps({
        "metric": "input MB/s tokenized (whole node), host buffers in -> ids in host buffers (PCIe and gather inclusive; tk_node_encode_batch_pinned)",
        "value": round(len(data) / 1e6 / (ms * 1e-3), 1), "unit": "MB/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
        "config": {"workload": "%d x %s documents per GPU, one process, tk_node_* over GPUs 0..%d, pinned caller buffers" % (args.docs, args.kind, n - 1),
                   "entry": "node", "docs_total": n_docs, "input_bytes_total": int(len(data)), "ids_total": int(n_ids)},
        "node": {"kernels_ms_max": round(float(np.mean(k_ms)), 4), "gather_ms": round(float(np.mean(g_ms)), 4)},
        "bit_exact_vs_cpu_sample_docs": sample, "bit_exact_vs_cpu": exact, "build": build_info()}), flush=True)
    node.close()
    if not exact:
        raise SystemExit(1)


def host_leg(args, tk, eng, data, offs, h_ids, h_oo):
    """the same batch from HOST memory to ids in HOST memory (PCIe inclusive; never `value`):
    tk_encode_batch_pipelined with pinned caller buffers (copy up / kernels / copy down overlapped, 32 MiB slices) next
    to the plain tk_encode_batch (pageable input, one copy up, kernels, one copy down)."""
    n_docs, n_bytes = len(offs) - 1, int(offs[-1])
    pin_data = tk.host_empty(n_bytes, np.uint8)
    pin_data[:] = data
    pin_offs = tk.host_empty(n_docs + 1, np.uint64)
    pin_offs[:] = offs
    pin_ids = tk.host_empty(n_bytes + 2 * n_docs + 1, np.uint32)
    pin_oo = tk.host_empty(n_docs + 1, np.uint64)
    t_pipe = []
    for it in range(args.host_steps + 1):
        t0 = time.perf_counter()
        ids, oo = eng.encode_batch_pipelined(pin_data, pin_offs, True, True, 0, pin_ids, pin_oo)
        if it:
            t_pipe.append(time.perf_counter() - t0)
    exact = bool(len(ids) == len(h_ids) and np.array_equal(ids, h_ids) and np.array_equal(oo, h_oo))
    t0 = time.perf_counter()
    eng.encode_batch(data, offs, True, True)
    t_plain = time.perf_counter() - t0
    tp = float(np.median(t_pipe))
    return {"metric": "input MB/s, host buffers in -> ids in host buffers (PCIe inclusive)", "pipelined_MBps": round(n_bytes / 1e6 / tp, 1),
            "pipelined_ms": round(tp * 1e3, 3), "plain_tk_encode_batch_MBps": round(n_bytes / 1e6 / t_plain, 1),
            "plain_ms": round(t_plain * 1e3, 3), "slice_bytes": 32 << 20, "buffers": "pinned (tk_host_alloc)", "bit_exact_vs_device_path": exact}


def decode_leg(args, tk, eng, v_ids, v_oo, n_docs, n_bytes, d_bytes, stream):
    """batch Tekkenizer::decode of the ids just produced, ids resident in HBM.
    Algorithmic bytes: 4 B per id read + 1 B per text byte written + two u64 offset arrays."""
    import torch
    ids = torch.as_tensor(v_ids, device="cuda").clone()
    oo = torch.as_tensor(v_oo, device="cuda").clone()
    ms = []
    for it in range(args.decode_steps + 1):
        v_b, v_bo = eng.decode_batch_device(ids.data_ptr(), oo.data_ptr(), n_docs, ids.numel(), tk.SpecialTokenPolicy.Ignore, stream)
        if it:
            ms.append(eng.last_timing()["pipeline_ms"])
    out = torch.as_tensor(v_b, device="cuda")
    ok = bool(out.numel() == n_bytes and torch.equal(out, d_bytes))
    t = float(np.mean(ms)) * 1e-3
    alg = 4 * ids.numel() + n_bytes + 16 * (n_docs + 1)
    return {"metric": "output MB/s decoded (pipeline: doclen + scan + emit + validate)", "value": round(n_bytes / 1e6 / t, 1),
            "ms": round(t * 1e3, 4), "round_trip_exact": ok,
            "roofline": {"bound": "hbm", "achieved": round(alg / t / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(alg / t / 1e9 / HBM_PEAK_GBS, 5)}}


def single_doc_leg(args, tk, eng, vocab_path):
    """The reference's OWN call shape: one &str per call (src/tekkenizer.rs:378-405).  BASELINE configs[0] shape: 1 000 x
    64-byte ASCII strings, one tk_encode_one call each, sequentially, timed by a C loop (tools/single_doc_bench.c: no Python
    in the timed region), next to the CPU oracle's time per document on the same strings (kind "port")."""
    import ctypes
    import subprocess
    import corpus
    import synth_vocab as sv
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import tk_oracle
    n = args.single_docs
    data, offs = corpus.generate("ascii", n, 64, seed=corpus.BASE_SEED)
    so = os.path.join(ROOT, "tools", "libtk_single_doc_bench.so")
    src = os.path.join(ROOT, "tools", "single_doc_bench.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        tmp = "%s.tmp%d" % (so, os.getpid())
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-o", tmp, src])
        os.replace(tmp, so)
    B = ctypes.CDLL(so)
    B.tkb_single_doc_loop.restype = ctypes.c_int
    B.tkb_single_doc_loop.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64, ctypes.c_int,
                                      ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    fn = ctypes.cast(tk.lib().tk_encode_one, ctypes.c_void_p)
    per_call = np.zeros(n, np.float64)
    fnv = ctypes.c_uint64(0)
    tot = ctypes.c_uint64(0)
    before = eng.small_path_calls()
    rc = B.tkb_single_doc_loop(fn, eng._h, data.ctypes.data, offs.ctypes.data, n, 3, per_call.ctypes.data, ctypes.byref(fnv), ctypes.byref(tot))
    if rc != 0:
        return {"error": "tk_encode_one failed with %d" % rc}
    one_launch = eng.small_path_calls() - before
    toks, ns, bos, eos = sv.load_tokens(vocab_path)
    orc = tk_oracle.Oracle(toks, ns, bos, eos)
    best = 1e9
    for _ in range(5):
        eids, _ = orc.encode_batch(data, offs, True, True, threads=1)
        best = min(best, tk_oracle.last_batch_seconds())
    us = per_call * 1e6
    return {"metric": "microseconds per Tekkenizer::encode call, one 64-byte document per call (BASELINE configs[0] shape), sequential",
            "calls": n, "us_per_call_median": round(float(np.median(us)), 2), "us_per_call_mean": round(float(np.mean(us)), 2),
            "us_per_call_p99": round(float(np.percentile(us, 99)), 2), "one_launch_calls": int(one_launch) // 4,
            "cpu_oracle_us_per_doc": round(best / n * 1e6, 2), "cpu_kind": "port (oracle/tk_oracle.c, 1 thread, no per-call allocation)",
            "bit_exact_vs_cpu": bool(fnv.value == tk_oracle.fnv1a(eids) and tot.value == len(eids)),
            "entry": "tk_encode_one (caller-owned output; text <= 64 KiB: one kernel launch, text and ids through mapped pinned memory)"}


def cpu_baseline(args, data, offs, vocab_path, ids, oo, n_bytes):
    """The oracle, single thread, on the same documents (the checker timed as the CPU baseline) and the
    bit-exact comparison of the GPU ids with it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import synth_vocab as sv
    import tk_oracle
    toks, ns, bos, eos = sv.load_tokens(vocab_path)
    orc = tk_oracle.Oracle(toks, ns, bos, eos)
    m = min(args.cpu_sample_docs, len(offs) - 1)
    sub, sub_offs = data[:int(offs[m])], offs[:m + 1]
    dt = 0.0
    for _ in range(args.cpu_passes):
        eids, eoo = orc.encode_batch(sub, sub_offs, True, True, threads=1)
        dt += tk_oracle.last_batch_seconds() / args.cpu_passes       # the C loop alone
    cpu_mbs = int(offs[m]) / 1e6 / dt
    exact = bool(np.array_equal(oo[:m + 1], eoo) and np.array_equal(ids[:int(oo[m])], eids))
    # what the vocabulary does to this corpus: the share of pieces that are not vocabulary keys is the merge kernels' workload
    k = min(m, 20000)
    st = orc.miss_stats(data[:int(offs[k])], offs[:k + 1])
    nt = {"vocab_fit": {"sample_docs": k, "pieces": st["pieces"], "miss_rate": round(st["missed"] / max(1, st["pieces"]), 4),
                        "mean_missed_piece_bytes": round(st["missed_bytes"] / max(1, st["missed"]), 2),
                        "mean_ids_per_missed_piece": round(st["missed_ids"] / max(1, st["missed"]), 2),
                        "bytes_per_piece": round(int(offs[k]) / max(1, st["pieces"]), 2)}}
    threads = args.cpu_threads or min(os.cpu_count() or 1, 256)
    if threads > 1:
        # BASELINE.md row CPU-restate-NT: the same restatement on N threads (documents striped), N stated
        orc.encode_batch(sub, sub_offs, True, True, threads=threads)     # (first pass: page faults of the staging buffer)
        orc.encode_batch(sub, sub_offs, True, True, threads=threads)
        dtn = tk_oracle.last_batch_seconds()
        nt["cpu_baseline_nt"] = {"value": round(int(offs[m]) / 1e6 / dtn, 1), "unit": "MB/s", "cores": threads, "kind": "port",
                                  "sample": "the same %d docs, second of 2 passes: %.3f s, oracle/tk_oracle.c on %d threads; host has %d cores"
                                            % (m, dtn, threads, os.cpu_count() or 0)}
    return dict(nt, **{"cpu_baseline": {"value": round(cpu_mbs, 1), "unit": "MB/s", "cores": 1, "kind": "port",
                             "sample": "%d docs (%d bytes) of the same workload, %d passes of %.1f s, oracle/tk_oracle.c single thread; host has %d cores"
                                       % (m, int(offs[m]), args.cpu_passes, dt, os.cpu_count() or 0)},
            "bit_exact_vs_cpu": exact, "fnv1a_ids": "%016x" % tk_oracle.fnv1a(ids)})


if __name__ == "__main__":
    main()
