"""tests/test_gpu_sharded.py::test_node_n_devices_on_one_gpu: the native N > 1 branch of csrc/tk_node.cpp -- worker threads, byte-balanced
runs, 18-bit packing, per-peer staging, unpacking, offset rebase by run -- with N contexts on ONE GPU and device-to-device copies
standing in for ncclSend / ncclRecv (TK_NODE_TRANSPORT=d2d, compiled into the development build only: `make ablate`).  The library is
chosen through TK_HIP_LIB before the package is imported, hence a process of its own.

  python tests/node_d2d_worker.py <out_path>"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    out_path = sys.argv[1]
    assert os.environ.get("TK_NODE_TRANSPORT") == "d2d" and "ablate" in os.environ.get("TK_HIP_LIB", "")
    import corpus
    import helpers
    tk = importlib.import_module("tekken-rs_amd")
    verdict = "ok"
    try:
        v = helpers.small_trained_vocab()
        orc = helpers.oracle_for(v)
        cases = []
        d, o = corpus.generate("ascii", 3000, 512, seed=corpus.BASE_SEED + 3)
        cases.append(("ascii", d, o))
        d, o = corpus.generate("zipf", 1500, seed=corpus.BASE_SEED + 4)
        cases.append(("zipf", d, o))
        # one document outweighs everything else: with 8 runs most of them are EMPTY (and one run holds empty documents only)
        docs = [b"tiny", b"", b"x" * 20000 + b" tail of the long one", b"", b"", b"end"]
        cases.append(("lopsided", *helpers_pack(docs)))
        cases.append(("two docs", *helpers_pack([b"hello world", "héllo".encode()])))
        cases.append(("no docs", np.zeros(1, np.uint8), np.zeros(1, np.uint64)))
        for n in (2, 3, 8):
            node = tk.Node(v["tokens"], v["num_special"], v["bos"], v["eos"], devices=(0,) * n)
            try:
                assert node.n_devices() == n
                for name, data, offs in cases:
                    for bos, eos in ((True, True), (False, False)):
                        ids, oo = node.encode_batch(data, offs, bos, eos)
                        eids, eoo = orc.encode_batch(data, offs, bos, eos, threads=8)
                        assert np.array_equal(oo, eoo) and np.array_equal(ids, eids), "N = %d, %s: ids differ from the oracle" % (n, name)
                    sb, si = node.last_shards()
                    lens = np.diff(offs.astype(np.int64))
                    total = int(offs[-1])
                    assert sum(sb) == total and sum(si) == len(ids), (n, name, sb, si)
                    longest = int(lens.max()) if len(lens) else 0
                    assert max(sb) <= total / n + longest + 1, "N = %d, %s: runs are not byte-balanced: %r" % (n, name, sb)
                    if name == "lopsided" and n == 8:
                        assert sum(1 for b in sb if b == 0) >= 5, sb
                # the caller-owned form through the same branch
                name, data, offs = cases[1]
                h_d, h_o = tk.host_empty(len(data), np.uint8), tk.host_empty(len(offs), np.uint64)
                h_d[:] = data
                h_o[:] = offs
                h_i, h_oo = tk.host_empty(len(data) + 2 * len(offs), np.uint32), tk.host_empty(len(offs), np.uint64)
                k = node.encode_batch_into(h_d, h_o, h_i, h_oo, True, True)
                eids, eoo = orc.encode_batch(data, offs, True, True, threads=8)
                assert k == len(eids) and np.array_equal(h_i[:k], eids) and np.array_equal(h_oo, eoo)
            finally:
                node.close()
    except Exception as e:  # noqa: BLE001
        import traceback
        verdict = "".join(traceback.format_exception(type(e), e, e.__traceback__))[-1500:]
    with open(out_path, "w") as f:
        f.write(verdict)


def helpers_pack(docs):
    offs = np.zeros(len(docs) + 1, np.uint64)
    offs[1:] = np.cumsum([len(d) for d in docs], dtype=np.uint64)
    return np.frombuffer(b"".join(docs) or b"\0", dtype=np.uint8).copy(), offs


if __name__ == "__main__":
    main()
