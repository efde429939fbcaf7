"""One-off differential fuzz of the GPU batch DECODE path (row f-1) against the restatement of Tekkenizer::decode
(oracle/tk_oracle.py decode_ref; test infrastructure): random batches of id lists -- empty documents anywhere, documents around
the 64-id step and the 16-document group boundaries, long documents, special ids under every policy, byte tokens that cut
UTF-8 sequences, ids outside the vocabulary -- through the pipeline's two length passes (groups of 16 documents and
TK_DECODE_GROUPS=0).  A batch the reference rejects must be rejected with the same error class and first bad document.
    python tools/gpu_fuzz_decode.py [--seconds 120] [--seed 1]"""
import argparse
import importlib
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

SPECIALS = ["<unk>", "<s>", "</s>", "[INST]", "[/INST]", "é\U0001f680"] + ["<SPECIAL_%d>" % i for i in range(6, 1000)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    import helpers
    import tk_oracle
    tk = importlib.import_module("tekken-rs_amd")
    v = helpers.small_trained_vocab()
    orc = tk_oracle.Oracle(v["tokens"], v["num_special"], v["bos"], v["eos"])
    ns, nr = v["num_special"], len(v["tokens"])
    engines = []
    for groups in ("1", "0"):
        os.environ["TK_DECODE_GROUPS"] = groups
        e = tk.Engine(v["tokens"], ns, v["bos"], v["eos"], device=0)
        e.set_special_tokens(SPECIALS)
        engines.append(e)
    rng = random.Random(a.seed)
    words = [b"hello", b" world", b"\n", b" caf\xc3\xa9", b" \xf0\x9f\x9a\x80", b"12", b" x", b"\t", b" the", b"ing", b" \xe4\xb8\xad\xe6\x96\x87"]

    def ref(ids, policy):
        return tk_oracle.decode_ref(v["tokens"], SPECIALS, ns, ids, policy)

    def one_doc(p_bad):
        k = rng.choice([0, 0, 1, 2, 10, 30, 63, 64, 65, 127, 128, 129, 400, 3000])
        ids = orc.encode(b"".join(rng.choice(words) for _ in range(k)), rng.random() < 0.5, rng.random() < 0.5)
        r = rng.random()
        if r < 0.25 and ids:                                    # special ids somewhere (between whole tokens: they may still cut a byte-fallback sequence)
            for _ in range(rng.randint(1, 3)):
                ids.insert(rng.randint(0, len(ids)), rng.randint(0, 5))
        elif r < 0.25 + 0.7 * p_bad and ids:                    # a raw byte token: may cut or break a UTF-8 sequence
            ids.insert(rng.randint(0, len(ids)), ns + rng.randint(0x80, 0xFF))
        elif r < 0.25 + p_bad:                                  # an id outside the vocabulary
            ids.insert(rng.randint(0, len(ids)), ns + nr + rng.randint(0, 9))
        return ids

    t0 = time.time()
    n_batches = n_docs = n_ids = n_err = 0
    while time.time() - t0 < a.seconds:
        nd = rng.choice([1, 2, 15, 16, 17, 31, 32, 33, 48, 100, 257])
        p_bad = rng.choice([0.0, 0.0, 0.0, 0.02, 0.2])              # most batches hold no document the reference rejects on purpose
        id_lists = [one_doc(p_bad) for _ in range(nd)]
        if rng.random() < 0.2:                                  # runs of empty documents: whole groups of them, the batch's end
            at = rng.randint(0, len(id_lists))
            id_lists[at:at] = [[] for _ in range(rng.choice([1, 16, 17, 40]))]
        policy = rng.choice([0, 0, 1, 2])
        want, err = [], None
        for d, ids in enumerate(id_lists):
            try:
                want.append(ref(ids, policy))
            except ValueError as ex:
                err = (d, str(ex))
                break
        for e in engines:
            if err is None:
                got = e.decode_docs(id_lists, policy)
                if got != want:
                    bad = next(i for i in range(len(want)) if got[i] != want[i])
                    raise SystemExit("MISMATCH seed %d batch %d doc %d policy %d: ids %r\n got %r\nwant %r" % (a.seed, n_batches, bad, policy, id_lists[bad][:80], got[bad][:200], want[bad][:200]))
            else:
                try:
                    e.decode_docs(id_lists, policy)
                except tk.TokenizerError as te:
                    kind = "SpecialTokenPolicy" if err[1] == "special" else "Tokenizers"
                    if te.kind != kind or te.bad_doc != err[0]:
                        raise SystemExit("ERROR CLASS seed %d batch %d: reference %r, GPU kind %s bad_doc %s" % (a.seed, n_batches, err, te.kind, te.bad_doc))
                else:
                    raise SystemExit("MISSED ERROR seed %d batch %d: reference %r, GPU returned text" % (a.seed, n_batches, err))
        n_err += err is not None
        n_batches += 1
        n_docs += len(id_lists)
        n_ids += sum(map(len, id_lists))
    for e in engines:
        e.close()
    print("decode fuzz ok: %d batches (%d rejected alike), %d documents, %d ids, both length passes (seed %d)" % (n_batches, n_err, n_docs, n_ids, a.seed))


if __name__ == "__main__":
    main()
