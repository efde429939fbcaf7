"""Throughput of the opt-in JSON pattern (TK_PIPELINE=doc times its sequential piece-by-piece path alone) next to the
default pipeline, on a sample of the C2 shape, with the oracle in the same mode as the checker.  Prints one JSON line."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")]


def main():
    import corpus
    import synth_vocab as sv
    import tk_oracle
    tk = importlib.import_module("tekken-rs_amd")
    toks, ns, bos, eos = sv.load_tokens(sv.ensure_default())
    eng = tk.Engine(toks, ns, bos, eos, device=0)
    n_docs = int(os.environ.get("N_DOCS", "200000"))
    kind, doc_len = os.environ.get("KIND", "ascii"), int(os.environ.get("DOC_LEN", "512"))
    data, offs = corpus.generate(kind, n_docs, doc_len, seed=corpus.BASE_SEED + (1 if kind == "ascii" else 2))
    import torch
    d_bytes = torch.from_numpy(data).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    out = {"what": "opt-in JSON pattern vs default", "kind": kind, "docs": n_docs, "bytes": int(offs[-1])}
    for mode in (0, 1):
        eng.set_pattern(mode)
        ms = []
        for it in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            v_ids, v_oo = eng.encode_batch_device_views(d_bytes.data_ptr(), d_offs.data_ptr(), n_docs, int(offs[-1]), True, True, 0)
            torch.cuda.synchronize()
            if it:
                ms.append((time.perf_counter() - t0) * 1e3)
        ids = torch.as_tensor(v_ids, device="cuda").cpu().numpy().view(np.uint32)
        oo = torch.as_tensor(v_oo, device="cuda").cpu().numpy().astype(np.uint64)
        orc = tk_oracle.Oracle(toks, ns, bos, eos)
        orc.set_pattern(mode)
        m = 5000
        t0 = time.perf_counter()
        eids, eoo = orc.encode_batch(data[:int(offs[m])], offs[:m + 1], True, True, threads=1)
        cpu_s = time.perf_counter() - t0
        key = "json_pattern" if mode else "default"
        handed = eng.last_stats()["handed_back"]
        out[key] = {"ms": round(float(np.median(ms)), 3), "MBps": round(int(offs[-1]) / 1e6 / (float(np.median(ms)) * 1e-3), 1),
                    "ids": int(len(ids)), "handed_back_docs": handed, "bit_exact_vs_oracle_sample": bool(np.array_equal(oo[:m + 1], eoo) and np.array_equal(ids[:int(oo[m])], eids)),
                    "cpu_oracle_MBps_1_thread": round(int(offs[m]) / 1e6 / cpu_s, 1)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
