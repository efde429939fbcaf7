#!/usr/bin/env python3
"""What the missed pieces of a bench corpus look like (CPU, oracle as the reader): how many ids a missed piece of each length
produces (what a tighter slot reservation than `len` would overflow on) and how often a missed piece of a FRESH batch was
already seen in earlier batches (the hit rate a memo table of merged pieces can have).
  python tools/miss_analysis.py --kind ascii --docs 100000 [--vocab-fit heldout] [--batches 4]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import corpus            # noqa: E402
import synth_vocab as sv  # noqa: E402
import tk_oracle          # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", default="ascii")
    ap.add_argument("--docs", type=int, default=100000)
    ap.add_argument("--doc-len", type=int, default=512)
    ap.add_argument("--vocab-fit", default="same")
    ap.add_argument("--batches", type=int, default=4)
    a = ap.parse_args()
    vp = sv.ensure_heldout() if a.vocab_fit == "heldout" else sv.ensure_default()
    toks, ns, bos, eos = sv.load_tokens(vp)
    orc = tk_oracle.Oracle(toks, ns, bos, eos)
    seen = np.zeros(0, np.uint64)
    for b in range(a.batches):
        data, offs = corpus.generate(a.kind, a.docs, a.doc_len, seed=corpus.BASE_SEED + 1 + 100 * b)
        rec = orc.miss_records(data, offs)
        ln, ni = rec[:, 1].astype(np.int64), rec[:, 2].astype(np.int64)
        key = (rec[:, 0].astype(np.uint64) << np.uint64(8)) | rec[:, 1].astype(np.uint64) & np.uint64(255)
        short = ln <= 16
        hit = np.isin(key, seen)
        print("batch %d: %d missed pieces, %.1f bytes, %.2f ids each; <= 16 bytes: %.3f; <= 16 bytes and <= 4 ids: %.3f; <= 3 ids %.3f" % (
            b, len(rec), ln.mean(), ni.mean(), short.mean(), (short & (ni <= 4)).mean(), (short & (ni <= 3)).mean()))
        print("   seen in earlier batches: %.4f of all, %.4f of the <= 16-byte ones; distinct keys so far %d" % (
            hit.mean(), hit[short].mean() if short.any() else 0.0, len(seen)))
        if b == 0:
            for name, res in (("len", ln), ("len/2+2", np.minimum(ln, ln // 2 + 2)), ("len/2+1", np.minimum(ln, ln // 2 + 1)), ("(len+3)/2", np.minimum(ln, (ln + 3) // 2))):
                over = ni > res
                print("   reservation %-10s: slots %.2f per piece, holes %.2f, overflow %.5f of the missed pieces" % (name, res.mean(), (res - ni)[~over].sum() / len(rec), over.mean()))
        seen = np.union1d(seen, key)


if __name__ == "__main__":
    main()
