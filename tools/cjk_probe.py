#!/usr/bin/env python3
"""How does the pipeline take text whose pieces are LONG by nature -- CJK paragraphs: runs of 5..60 ideographs (15..180
bytes, one \\p{L}+ piece each) between full-width punctuation?  Pieces over 64 bytes send their document to the
per-document kernels (pass 2).  Run on the GPU box:  python tools/cjk_probe.py [n_docs]"""
import importlib
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import synth_vocab as sv  # noqa: E402


def make_docs(n_docs, doc_bytes, max_run, seed):
    rng = random.Random(seed)
    common = [chr(0x4E00 + rng.randrange(0x5000)) for _ in range(3000)]
    weights = [1.0 / (i + 1) for i in range(len(common))]
    docs = []
    for _ in range(n_docs):
        parts, size = [], 0
        while size < doc_bytes:
            run = "".join(rng.choices(common, weights, k=rng.randint(5, max_run)))
            parts.append(run + rng.choice("，。、；"))
            size += 3 * len(run) + 3
        docs.append("".join(parts).encode()[:doc_bytes // 3 * 3])
    return docs


def main():
    n_docs = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    tk = importlib.import_module("tekken-rs_amd")
    import tk_oracle
    toks, ns, bos, eos = sv.load_tokens(sv.ensure_default())
    e = tk.Engine(toks, ns, bos, eos, device=0)
    orc = tk_oracle.Oracle(toks, ns, bos, eos)
    for max_run in (15, 21, 30, 60):
        docs = make_docs(n_docs, 2048, max_run, 7 + max_run)
        data, offs = tk.pack_docs(docs)
        best = 1e9
        for _ in range(4):
            ids, oo = e.encode_batch(data, offs, True, True)
            best = min(best, e.last_timing()["pipeline_ms"])
        t = e.last_stats()
        sample = 2000
        eids, eoo = orc.encode_batch(data[:int(offs[sample])], offs[:sample + 1], True, True, threads=8)
        ok = bool(np.array_equal(ids[:int(oo[sample])], eids))
        print("runs of 5..%d ideographs (<= %d bytes): %d docs, %.1f MB, pipeline %.2f ms = %.1f GB/s, handed-back docs %s, long-piece records %d, regions through the cut path %d, bit-exact on %d docs: %s"
              % (max_run, 3 * max_run, n_docs, len(data) / 1e6, best, len(data) / best / 1e6, t["handed_back"], e.long_piece_records(), e.cut_chunks(), sample, ok), flush=True)
    e.close()


if __name__ == "__main__":
    main()
