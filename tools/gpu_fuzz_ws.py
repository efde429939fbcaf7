"""One-off differential fuzz (test infrastructure) aimed at where round 4's split bug showed: LONG pieces of mixed white space
(blanks, TABs, CR / LF, U+00A0, U+3000, U+2009) and of punctuation with line ends (40..400 bytes: cut into fragments across chunk
boundaries), between short words, digits and multi-byte chars, at random alignments; both patterns.
    python tools/gpu_fuzz_ws.py [--seconds 120] [--seed 1] [--pattern 0|1]"""
import argparse
import importlib
import os
import random
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--pattern", type=int, choices=[0, 1], default=0)
    a = ap.parse_args()
    import helpers
    import tk_oracle
    tk = importlib.import_module("tekken-rs_amd")
    v = helpers.small_trained_vocab()
    orc = tk_oracle.Oracle(v["tokens"], v["num_special"], v["bos"], v["eos"])
    orc.set_pattern(a.pattern)
    eng = tk.Engine(v["tokens"], v["num_special"], v["bos"], v["eos"], device=0)
    eng.set_pattern(a.pattern)
    rng = random.Random(a.seed)
    ws = [" ", "\t", "\r", "\n", " ", "　", " ", "\r\n"]
    punct = ["!", "-", ".", "/", "'", "…", "¿", "́"]
    words = ["a", "Ab", "the", "1", "23", "中", "été", "\U0001f680", "x" * 30, "٣٣"]

    def long_run():
        kind = rng.random()
        out = []
        n = rng.choice([40, 64, 65, 100, 136, 200, 400])
        if kind < 0.6:                      # white space: a few kinds, in stretches
            pool = rng.sample(ws, rng.randint(1, 4))
            while sum(map(len, out)) < n:
                out.append(rng.choice(pool) * rng.choice([1, 2, 11, 22, 40]))
        else:                               # punctuation with line ends / slashes inside and behind
            pool = rng.sample(punct, rng.randint(1, 3)) + rng.sample(["\r", "\n", "/"], rng.randint(0, 2))
            while sum(map(len, out)) < n:
                out.append(rng.choice(pool) * rng.choice([1, 3, 11, 31, 40]))
        return "".join(out)

    t0 = time.time()
    it = n_docs = n_bytes = 0
    while time.time() - t0 < a.seconds:
        docs = []
        for _ in range(rng.randint(1, 30)):
            parts = []
            for _ in range(rng.randint(0, 25)):
                parts.append(long_run() if rng.random() < 0.35 else rng.choice(words) + rng.choice(["", " ", "\n", ", "]))
            docs.append(("".join(parts))[:60000].encode())
        bos, eos = rng.random() < 0.5, rng.random() < 0.5
        data, offs = tk.pack_docs(docs)
        ids, oo = eng.encode_batch(data, offs, bos, eos)
        eids, eoo = orc.encode_batch(data, offs, bos, eos, threads=8)
        if not (np.array_equal(oo, eoo) and np.array_equal(ids, eids)):
            for d in range(len(docs)):
                if ids[int(oo[d]):int(oo[d + 1])].tolist() != eids[int(eoo[d]):int(eoo[d + 1])].tolist():
                    print("MISMATCH seed", a.seed, "iteration", it, "doc", d, "of", len(docs), "len", len(docs[d]))
                    import pickle
                    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
                    pickle.dump((docs, bos, eos, a.pattern), open(os.path.join(ROOT, "gpurun_out", "fuzz_ws_fail.pkl"), "wb"))
                    sys.exit(1)
            print("offset arrays differ")
            sys.exit(1)
        it += 1
        n_docs += len(docs)
        n_bytes += len(data)
    print("ws fuzz ok: %d batches, %d documents, %d bytes (seed %d, pattern %d), handed back in the last batch: %s" % (it, n_docs, n_bytes, a.seed, a.pattern, eng.last_stats()))


if __name__ == "__main__":
    main()
