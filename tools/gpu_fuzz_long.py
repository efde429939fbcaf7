#!/usr/bin/env python3
"""Differential fuzz of the HIP path against the oracle on the GPU box: random documents over a small alphabet that is rich
in rule boundaries (letters, digits, apostrophes, white space of several kinds, CR / LF, multi-byte letters, marks, emoji), with
runs of every length up to a few hundred bytes so that every piece class occurs -- lookup hits, the merge kernels' 8 / 16 /
32 / 64-entry columns, the long-piece records (65..128: one lane per piece; 129..256: one wave, parts in registers), pieces
beyond 256 bytes (handed back), chunk boundaries inside all of them.  python tools/gpu_fuzz_long.py [seconds] [seed]   (tools/gpu_fuzz.py is the general one)"""
import importlib
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import synth_vocab as sv  # noqa: E402
import helpers  # noqa: E402
import tk_oracle  # noqa: E402

ALPHA = ["a", "b", "S", "s", "t", "r", "e", "l", "v", "m", "d", "x", "1", "2", "9", "'", "!", "=", "-", " ", " ", "\n", "\r", "\t",
         "ſ", "é", "中", "文", "字", "٣", " ", " ", "　", "\U0001f680", "́", "，", "。"]


def make_doc(rng):
    parts = []
    for _ in range(rng.randint(0, rng.choice([3, 12, 60, 400]))):
        c = rng.choice(ALPHA)
        k = rng.choice([1, 1, 1, 1, 2, 3, 5, 9, 17, 33, 65, 100, 129, 200, 257, 300, 700, 2100, 4500])   # (beyond 64 bytes: the cut path)
        if k > 3:
            k = rng.randint(k // 2 + 1, k)
        u = rng.random()
        if u < 0.4:
            parts.append(c * k)
        elif u < 0.6:
            parts.append("".join(rng.choice("abcdefghijklmnopqrstuvwxyz") for _ in range(k)))   # random letters: cut into fragments
        elif u < 0.7:
            parts.append("".join(rng.choice("ab") for _ in range(k)))                           # few distinct pairs: hardly any cut
        else:
            parts.append("".join(rng.choice(ALPHA[:12] + ["中", "文", "é"]) for _ in range(k)))
    return "".join(parts).encode("utf-8")


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    tk = importlib.import_module("tekken-rs_amd")
    toks, ns, bos, eos = sv.load_tokens(sv.ensure_default())
    small = helpers.small_trained_vocab()
    engines = [(tk.Engine(toks, ns, bos, eos, device=0), tk_oracle.Oracle(toks, ns, bos, eos), "bench vocabulary"),
               (tk.Engine(small["tokens"], small["num_special"], small["bos"], small["eos"], device=0), helpers.oracle_for(small), "small vocabulary")]
    rng = random.Random(seed)
    t0 = time.time()
    batches = docs_total = bytes_total = recs = handed = cuts = late = 0
    while time.time() - t0 < seconds:
        docs = [make_doc(rng) for _ in range(rng.choice([1, 7, 300, 3000]))]
        data, offs = tk.pack_docs(docs)
        bosf, eosf = rng.random() < 0.5, rng.random() < 0.5
        for eng, orc, name in engines:
            ids, oo = eng.encode_batch(data, offs, bosf, eosf)
            eids, eoo = orc.encode_batch(data, offs, bosf, eosf, threads=16)
            if not (np.array_equal(oo, eoo) and np.array_equal(ids, eids)):
                for d in range(len(docs)):
                    a, b = ids[int(oo[d]):int(oo[d + 1])], eids[int(eoo[d]):int(eoo[d + 1])]
                    if a.tolist() != b.tolist():
                        print("MISMATCH (%s) seed %d batch %d doc %d: %r" % (name, seed, batches, d, docs[d][:300]), flush=True)
                        sys.exit(1)
            recs += eng.long_piece_records()
            cuts += eng.cut_chunks()
            late += 1 if eng.last_host_syncs() > 2 else 0
            handed += eng.last_stats()["handed_back"]
        batches += 1
        docs_total += len(docs)
        bytes_total += len(data)
    print("gpu_fuzz_long: %d batches, %d documents, %.1f MB, both vocabularies: bit-exact; %d long-piece records, %d regions through the cut path, %d documents handed back, %d calls with late-flagged documents (seed %d, %.0f s)"
          % (batches, docs_total, bytes_total / 1e6, recs, cuts, handed, late, seed, time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
