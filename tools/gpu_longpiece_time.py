#!/usr/bin/env python3
"""Time ONE long single-piece document through the pipeline: the round-based workgroup merge (csrc/tk_long.hip) against the
step-by-step single-wave merge (TK_LONG_MIN=0).  Run on the GPU box:  python tools/gpu_longpiece_time.py"""
import importlib
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import synth_vocab as sv  # noqa: E402


def main():
    tk = importlib.import_module("tekken-rs_amd")
    toks, ns, bos, eos = sv.load_tokens(sv.ensure_default())
    rng = random.Random(3)
    cases = {"random letters": lambda n: bytes(rng.randrange(97, 123) for _ in range(n)), "one letter": lambda n: b"a" * n,
             "white space": lambda n: bytes(rng.choice(b" " * 38 + b"\n\t") for _ in range(n - 1)) + b"\n"}
    for lm in ("1024", "0"):
        os.environ["TK_LONG_MIN"] = lm
        e = tk.Engine(toks, ns, bos, eos, device=0)
        for name, gen in cases.items():
            for n in (2048, 8192, 32768):
                doc = gen(n)
                data, offs = tk.pack_docs([doc])
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    ids, _ = e.encode_batch(data, offs, True, True)
                    best = min(best, time.perf_counter() - t0)
                print("TK_LONG_MIN=%-5s %-15s %6d bytes -> %6d ids  %.3f ms (wall), pipeline %.3f ms, round-path docs so far %d"
                      % (lm, name, n, len(ids), best * 1e3, e.last_timing()["pipeline_ms"], e.round_path_docs()), flush=True)
        e.close()


if __name__ == "__main__":
    main()
