#!/usr/bin/env python3
"""Experiment: how much does a second batch in flight buy on ONE GPU?
Two contexts (own tables, scratch, streams), two host threads, the C2 shape either whole (each thread its own 1 M-document
batch) or split (each thread one half of a 1 M-document batch).  Prints ms per 512 MB of input for 1 and 2 threads.
Run on the GPU box: python tools/experiments/two_in_flight.py [--steps 100]"""
import argparse, importlib, os, sys, threading, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--docs", type=int, default=1_000_000)
    ap.add_argument("--kind", default="ascii")
    ap.add_argument("--doc-len", type=int, default=512)
    ap.add_argument("--vocab-fit", default="same")
    args = ap.parse_args()
    import torch, corpus, synth_vocab as sv
    tk = importlib.import_module("tekken-rs_amd")
    vp = sv.ensure_heldout() if args.vocab_fit == "heldout" else sv.ensure_default()
    toks = [tk.Tekkenizer.from_file(vp, device=0) for _ in range(2)]
    engs = [t.engine() for t in toks]
    for e in engs:
        e.set_memo(0, 0)
    data, offs = corpus.generate(args.kind, args.docs, args.doc_len, seed=corpus.BASE_SEED + 1)
    n_bytes = int(offs[-1])
    d_bytes = torch.from_numpy(data).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    half = args.docs // 2
    cut = int(offs[half])
    offs_b = (offs[half:] - cut).astype(np.int64)
    d_offs_b = torch.from_numpy(offs_b).cuda()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()

    def whole(i, k):
        for _ in range(k):
            engs[i].encode_batch_device_views(d_bytes.data_ptr(), d_offs.data_ptr(), args.docs, n_bytes, True, True, streams[i].cuda_stream)

    def halves(i, k):
        for _ in range(k):
            if i == 0:
                engs[0].encode_batch_device_views(d_bytes.data_ptr(), d_offs.data_ptr(), half, cut, True, True, streams[0].cuda_stream)
            else:
                engs[1].encode_batch_device_views(d_bytes.data_ptr() + cut, d_offs_b.data_ptr(), args.docs - half, n_bytes - cut, True, True, streams[1].cuda_stream)

    def timed(fn, nthreads, k):
        ths = [threading.Thread(target=fn, args=(i, k)) for i in range(nthreads)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    for fn in (whole, halves):
        timed(fn, 2, 3)
    k = args.steps
    t1 = timed(whole, 1, k) / k
    t2 = timed(whole, 2, k) / (2 * k)
    print("whole batches : 1 in flight %.4f ms / batch   2 in flight %.4f ms / batch  (x%.3f)" % (t1, t2, t1 / t2))
    h2 = timed(halves, 2, k) / k
    print("two halves of one batch, one context each, concurrently: %.4f ms / batch (x%.3f)" % (h2, t1 / h2))
    # the same two halves one after the other on one thread (what splitting alone costs)
    def serial_halves(i, k):
        for _ in range(k):
            engs[0].encode_batch_device_views(d_bytes.data_ptr(), d_offs.data_ptr(), half, cut, True, True, streams[0].cuda_stream)
            engs[1].encode_batch_device_views(d_bytes.data_ptr() + cut, d_offs_b.data_ptr(), args.docs - half, n_bytes - cut, True, True, streams[1].cuda_stream)
    hs = timed(serial_halves, 1, k) / k
    print("two halves one after the other: %.4f ms / batch" % hs)


if __name__ == "__main__":
    main()
