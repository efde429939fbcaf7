"""One-off differential fuzz of the GPU encode path against the oracle (test infrastructure): random batches of
documents over several alphabets / length mixes, both pipelines' paths (flat, hand-back, long pieces), BOS / EOS mixes.
    python tools/gpu_fuzz.py [--seconds 120] [--seed 1]"""
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
    ap.add_argument("--vocab", choices=["small", "bench"], default="small")
    ap.add_argument("--pattern", type=int, choices=[0, 1], default=0, help="1: the opt-in JSON pattern of tekken.json (row f-3)")
    a = ap.parse_args()
    import helpers
    tk = importlib.import_module("tekken-rs_amd")
    if a.vocab == "bench":
        import synth_vocab as sv
        toks, ns, bos, eos = sv.load_tokens(sv.ensure_default())
        v = {"tokens": toks, "num_special": ns, "bos": bos, "eos": eos}
    else:
        v = helpers.small_trained_vocab()
    import tk_oracle
    orc = tk_oracle.Oracle(v["tokens"], v["num_special"], v["bos"], v["eos"])
    orc.set_pattern(a.pattern)
    eng = tk.Engine(v["tokens"], v["num_special"], v["bos"], v["eos"], device=0)
    eng.set_pattern(a.pattern)
    rng = random.Random(a.seed)
    alphabets = [
        list("abcdefghijklmnopqrstuvwxyz") + [" "] * 8 + list(".,;!?'\n"),
        list("0123456789") * 3 + list(" ,.;:-+\n\t"),
        list("aA bB'sS tT!\n\r\t 12"),
        ["a", "S", "1", "٣", "３", "'", "ſ", "s", "!", " ", " ", "\n", "\r", "中", "é", "\U0001f680", " ", "　", "-", "\t"],
        list("xyz") + [" "],
        list("aAbBcCzZ") + [" "] * 4 + list("12/!-\n\r.") + ["é", "É", "Ж", "ж", "ǅ", "中", "文", "ʰ", "ª", "́", "̈", "\U0001f680", "¿", "—"],
        [chr(c) for c in range(32, 127)] + ["\n", "\t", "\r"],
    ]
    t0 = time.time()
    it = n_docs = n_bytes = 0
    while time.time() - t0 < a.seconds:
        docs = []
        target = rng.choice([2000, 20000, 200000])
        while sum(map(len, docs)) < target:
            al = rng.choice(alphabets)
            n = rng.choice([0, 1, 2, 5, 17, 60, 300, 900, 1900, 2100, 5000])
            n = rng.randint(0, n)
            rep = rng.choice([1, 1, 1, 1, 2, 4, 40, 100, 3000])
            s = "".join(rng.choice(al) * rng.randint(1, rep) for _ in range(n))[:40000]
            docs.append(s.encode("utf-8", "ignore"))
        bos, eos = rng.random() < 0.5, rng.random() < 0.5
        data, offs = tk.pack_docs(docs)
        if rng.random() < 0.03:
            eng.memo_clear()          # (the next call fills an empty memo table again: its own path in the merge kernel)
        ids, oo = eng.encode_batch(data, offs, bos, eos)
        eids, eoo = orc.encode_batch(data, offs, bos, eos, threads=8)
        if not (np.array_equal(oo, eoo) and np.array_equal(ids, eids)):
            for d in range(len(docs)):
                if ids[int(oo[d]):int(oo[d + 1])].tolist() != eids[int(eoo[d]):int(eoo[d + 1])].tolist():
                    print("MISMATCH iteration", it, "doc", d, "len", len(docs[d]), repr(docs[d][:200]))
                    open(os.path.join(ROOT, "gpurun_out", "fuzz_fail.bin"), "wb").write(docs[d])
                    sys.exit(1)
            print("offset arrays differ")
            sys.exit(1)
        it += 1
        n_docs += len(docs)
        n_bytes += len(data)
    print("fuzz ok: %d batches, %d documents, %d bytes, handed back in the last batch: %s" % (it, n_docs, n_bytes, eng.last_stats()))


if __name__ == "__main__":
    main()
