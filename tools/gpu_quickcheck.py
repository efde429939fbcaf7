#!/usr/bin/env python3
"""Quick GPU sanity run (dev tool): parity of the HIP path vs the oracle on small corpora and a
first timing on a slice of the C2 workload.  Usage: python tools/gpu_quickcheck.py [n_docs]"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import corpus  # noqa: E402
import synth_vocab as sv  # noqa: E402
import tk_oracle  # noqa: E402

tk = importlib.import_module("tekken-rs_amd")


def main():
    n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    toks, ns, bos, eos = sv.load_tokens(sv.ensure_default())
    t0 = time.time()
    eng = tk.Engine(toks, ns, bos, eos, device=0)
    print("ctx_create %.2fs" % (time.time() - t0), flush=True)
    orc = tk_oracle.Oracle(toks, ns, bos, eos)

    for kind, n, dl, sd in (("ascii", 2000, 512, 1), ("mixed", 500, 2048, 2), ("zipf", 300, 0, 4), ("zipf", 3000, 0, 4)):
        data, offs = corpus.generate(kind, n, dl, seed=corpus.BASE_SEED + sd)
        print("running", kind, n, "bytes", len(data), flush=True)
        t0 = time.time()
        ids, oo = eng.encode_batch(data, offs, True, True, validate_utf8=True)
        t1 = time.time()
        eids, eoo = orc.encode_batch(data, offs, True, True, threads=8)
        ok = np.array_equal(oo, eoo) and np.array_equal(ids, eids)
        print("%s: docs=%d bytes=%d ids=%d parity=%s gpu_call=%.3fs stats=%s timing=%s" %
              (kind, n, len(data), len(ids), ok, t1 - t0, eng.last_stats(), eng.last_timing()), flush=True)
        if not ok:
            for d in range(n):
                a = ids[int(oo[d]):int(oo[d + 1])]
                b = eids[int(eoo[d]):int(eoo[d + 1])]
                if len(a) != len(b) or not np.array_equal(a, b):
                    doc = data[int(offs[d]):int(offs[d + 1])].tobytes()
                    k = 0
                    while k < min(len(a), len(b)) and a[k] == b[k]:
                        k += 1
                    print("first bad doc", d, "len", len(doc), "token", k, a[max(0, k - 3):k + 4], b[max(0, k - 3):k + 4])
                    print(repr(doc[:200]))
                    break
            sys.exit(1)

    import torch
    data, offs = corpus.generate("ascii", n_docs, 512, seed=corpus.BASE_SEED + 1)
    d_bytes = torch.from_numpy(data).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    torch.cuda.synchronize()
    for it in range(4):
        t0 = time.time()
        p_ids, p_oo, n_ids = eng.encode_batch_device(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, len(data), True, True,
                                                      torch.cuda.current_stream().cuda_stream)
        dt = time.time() - t0
        tm = eng.last_timing()
        print("C2 slice: docs=%d bytes=%d ids=%d wall=%.2fms pipeline=%.2fms encode=%.2fms  -> %.1f MB/s (wall), %.1f MB/s (kernel)"
              % (n_docs, len(data), n_ids, dt * 1e3, tm["pipeline_ms"], tm["encode_kernel_ms"], len(data) / dt / 1e6,
                 len(data) / (tm["encode_kernel_ms"] * 1e-3) / 1e6), flush=True)
    # PCIe-inclusive rate of the host-buffer entry (pageable input, pinned output; never the bench `value`)
    for it in range(3):
        t0 = time.time()
        ids, oo = eng.encode_batch(data, offs, True, True)
        dt = time.time() - t0
        print("host entry tk_encode_batch (H2D + pipeline + D2H): %.1f ms -> %.1f MB/s PCIe-inclusive" % (dt * 1e3, len(data) / dt / 1e6), flush=True)
    t0 = time.time()
    sample = 20000
    eids, eoo = orc.encode_batch(data[:int(offs[sample])], offs[:sample + 1], True, True, threads=1)
    dt = time.time() - t0
    print("oracle 1 thread: %d docs in %.2fs = %.1f MB/s" % (sample, dt, int(offs[sample]) / dt / 1e6))


if __name__ == "__main__":
    main()
