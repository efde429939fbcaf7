"""Differential stress of pass 2 (one wave per long piece, csrc/tk_encode_impl.h tk_piece_coop) against the oracle
(test infrastructure): documents that are ONE piece of 0.5 .. 32 KiB -- random letters, few-letter alphabets, runs --
so that thousands of dependent merges go through the node arrays in global memory.
    python tools/gpu_longpiece_stress.py [--docs 160] [--seed 5] [--vocab small|bench]"""
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
    ap.add_argument("--docs", type=int, default=160)
    ap.add_argument("--seed", type=int, default=5)
    ap.add_argument("--vocab", choices=["small", "bench"], default="bench")
    a = ap.parse_args()
    import helpers
    import tk_oracle
    tk = importlib.import_module("tekken-rs_amd")
    if a.vocab == "bench":
        import synth_vocab as sv
        toks, ns, bos, eos = sv.load_tokens(sv.ensure_default())
        v = {"tokens": toks, "num_special": ns, "bos": bos, "eos": eos}
    else:
        v = helpers.small_trained_vocab()
    orc = tk_oracle.Oracle(v["tokens"], v["num_special"], v["bos"], v["eos"])
    eng = tk.Engine(v["tokens"], v["num_special"], v["bos"], v["eos"], device=0)
    rng = random.Random(a.seed)
    alphabets = ["abcdefghijklmnopqrstuvwxyz", "ab", "etaoinshr", "abcABC", "x"]
    docs = []
    for i in range(a.docs):
        n = rng.choice([512, 1000, 2047, 4096, 8191, 16384, 30000, 32768])
        al = rng.choice(alphabets)
        docs.append("".join(rng.choice(al) for _ in range(n)).encode())
    data, offs = tk.pack_docs(docs)
    t0 = time.time()
    eids, eoo = orc.encode_batch(data, offs, True, True, threads=16)
    t1 = time.time()
    for rep in range(3):
        ids, oo = eng.encode_batch(data, offs, True, True)
        assert np.array_equal(oo, eoo) and np.array_equal(ids, eids), "pass 2 differs from the oracle (rep %d)" % rep
    print("long-piece stress ok: %d documents, %d bytes, %d ids, oracle %.1f s, stats %s" % (
        len(docs), int(offs[-1]), len(ids), t1 - t0, eng.last_stats()))
    eng.close()


if __name__ == "__main__":
    main()
