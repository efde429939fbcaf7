#!/usr/bin/env python3
"""What would software pipelining of sub-batches inside one call buy on the C2 shape?  The cheap way
to find out before building it: K contexts, each with its own stream and workspace, each taking 1 / K of the 1 M x 512-byte
batch, called from K host threads at once (ctypes releases the GIL) -- the kernels of the slices then overlap exactly as
they would with K slices in flight inside one call (flat kernel of one slice beside merge / assembly of another), without
any of the plumbing.  Compared with ONE context over the whole batch, same box, same process.
    python tools/experiments/overlap_probe.py [steps]"""
import importlib
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import corpus  # noqa: E402
import synth_vocab as sv  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    tk = importlib.import_module("tekken-rs_amd")
    toks, ns, bos, eos = sv.load_tokens(sv.ensure_default())
    n_docs = 1_000_000
    data, offs = corpus.generate("ascii", n_docs, 512, seed=corpus.BASE_SEED + 1)
    d_bytes = torch.from_numpy(data).cuda()
    for K in (1, 2, 4):
        engines = [tk.Engine(toks, ns, bos, eos, device=0) for _ in range(K)]
        streams = [torch.cuda.Stream() for _ in range(K)]
        parts = []
        for k in range(K):
            lo, hi = n_docs * k // K, n_docs * (k + 1) // K
            o = (offs[lo:hi + 1] - offs[lo]).astype(np.int64)
            parts.append((int(offs[lo]), hi - lo, int(offs[hi] - offs[lo]), torch.from_numpy(o).cuda()))
        torch.cuda.synchronize()

        def work(k, n):
            base, nd, nb, d_o = parts[k]
            for _ in range(n):
                engines[k].encode_batch_device_views(d_bytes.data_ptr() + base, d_o.data_ptr(), nd, nb, True, True, streams[k].cuda_stream)

        for warm in (True, False):
            n = 5 if warm else steps
            th = [threading.Thread(target=work, args=(k, n)) for k in range(K)]
            torch.cuda.synchronize()
            t0 = time.time()
            for t in th:
                t.start()
            for t in th:
                t.join()
            torch.cuda.synchronize()
            dt = time.time() - t0
        print("K = %d contexts side by side: %.4f ms per whole batch (%.1f GB/s); kernel times of context 0: %s"
              % (K, dt / steps * 1e3, len(data) / (dt / steps) / 1e9, engines[0].last_timing()), flush=True)
        for e in engines:
            e.close()


if __name__ == "__main__":
    main()
