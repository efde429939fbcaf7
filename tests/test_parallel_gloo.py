"""The N>1 path (byte-balanced sharding + variable-length gather to rank 0) with world_size 2 and 3
on the gloo backend.  The per-rank encoder is the oracle here (this is a CPU test of the
distributed plumbing, tekken-rs_amd/parallel.py); on the GPU box the same code runs with the HIP
engine and backend nccl (= RCCL)."""
import importlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class NumpyIds18Codec:
    """The 18-bit wire format of include/tekken_hip.h (tk_ids18_bytes) on CPU tensors -- TEST infrastructure for the
    gather logic under gloo; the product codec is parallel.Ids18Codec (HIP kernels)."""

    @staticmethod
    def nbytes(n):
        return ((2 * n + 3) & ~3) + 4 * ((n + 15) // 16)

    def packed_numel(self, n):
        return (self.nbytes(n) + 3) // 4

    def pack(self, ids):
        import torch
        v = ids.numpy().view(np.uint32)
        n = len(v)
        assert n == 0 or int(v.max()) < (1 << 18)
        out = np.zeros(self.packed_numel(n) * 4, np.uint8)
        out[:2 * n] = (v & 0xFFFF).astype(np.uint16).view(np.uint8)
        hi = np.zeros(((n + 15) // 16) * 16, np.uint32)
        hi[:n] = (v >> 16) & 3
        words = (hi.reshape(-1, 16) << (2 * np.arange(16, dtype=np.uint32))).sum(axis=1).astype(np.uint32)
        h0 = (2 * n + 3) & ~3
        out[h0:h0 + 4 * len(words)] = words.view(np.uint8)
        return torch.from_numpy(out.view(np.int32).copy())

    def unpack(self, packed, n, out):
        raw = packed.numpy().view(np.uint8)
        lows = raw[:2 * n].view(np.uint16).astype(np.uint32)
        h0 = (2 * n + 3) & ~3
        words = raw[h0:h0 + 4 * ((n + 15) // 16)].view(np.uint32)
        hi = ((words[:, None] >> (2 * np.arange(16, dtype=np.uint32))) & 3).reshape(-1)[:n]
        out.numpy().view(np.uint32)[:] = lows | (hi << 16)


def _worker(rank, world, port, out_path):
    for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
        sys.path.insert(0, p)
    import torch.distributed as dist
    import corpus
    import helpers
    par = importlib.import_module("tekken-rs_amd.parallel")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        v = helpers.small_trained_vocab()
        orc = helpers.oracle_for(v)
        data, offs = corpus.generate("zipf", 400, seed=corpus.BASE_SEED + 4)

        def enc(d, o, bos, eos):
            return orc.encode_batch(d, o, bos, eos)

        ids, oo = par.encode_sharded(enc, data, offs, True, True, dst=0)
        if rank == 0:
            eids, eoo = orc.encode_batch(data, offs, True, True)
            ok = np.array_equal(ids, eids) and np.array_equal(oo, eoo)
            # empty shard edge: more ranks than documents
            with open(out_path, "w") as f:
                f.write("ok" if ok else "mismatch")
        # the 18-bit wire format and the non-blocking form (two gathers in flight, results taken in order)
        import torch
        cuts = par.shard_by_bytes(offs, world)
        d0, d1 = cuts[rank], cuts[rank + 1]
        lids, loo = orc.encode_batch(data[int(offs[d0]):int(offs[d1])], (offs[d0:d1 + 1] - offs[d0]).astype(np.uint64), True, True)
        t_ids = torch.from_numpy(np.ascontiguousarray(lids).view(np.int32))
        t_cnt = torch.from_numpy(np.diff(loo.astype(np.int64)))
        codec = NumpyIds18Codec()
        p1 = par.gather_ids(t_ids, t_cnt, dst=0, codec=codec, wait=False)
        p2 = par.gather_ids(t_ids.clone(), t_cnt.clone(), dst=0, codec=None, wait=False)
        for pend in (p1, p2):
            g_ids, g_offs = pend.result()
            if rank == 0:
                if not (np.array_equal(g_ids.numpy().view(np.uint32), eids) and np.array_equal(g_offs.numpy().astype(np.uint64), eoo)):
                    with open(out_path, "w") as f:
                        f.write("mismatch-codec")
        # a second round where one rank has nothing to send
        data2, offs2 = data[:int(offs[1])], offs[:2]
        ids2, oo2 = par.encode_sharded(enc, data2, offs2, False, False, dst=0)
        if rank == 0:
            e2, eo2 = orc.encode_batch(data2, offs2, False, False)
            if not (np.array_equal(ids2, e2) and np.array_equal(oo2, eo2)):
                with open(out_path, "w") as f:
                    f.write("mismatch-single-doc")
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_encode_gathers_in_document_order(world, tmp_path):
    import torch.multiprocessing as mp
    out = str(tmp_path / "result.txt")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert open(out).read() == "ok"


def test_shard_by_bytes_balances_bytes():
    par = importlib.import_module("tekken-rs_amd.parallel")
    rng = np.random.default_rng(0)
    lens = np.concatenate([rng.integers(16, 200, 1000), [30000, 30000], rng.integers(16, 200, 1000)])
    offs = np.zeros(len(lens) + 1, np.uint64)
    offs[1:] = np.cumsum(lens)
    for world in (1, 2, 4, 8):
        cuts = par.shard_by_bytes(offs, world)
        assert cuts[0] == 0 and cuts[-1] == len(lens) and all(a <= b for a, b in zip(cuts, cuts[1:]))
        per = [int(offs[cuts[r + 1]]) - int(offs[cuts[r]]) for r in range(world)]
        assert max(per) - min(per) <= 2 * 30000 + 200
    assert par.shard_by_bytes(np.zeros(1, np.uint64), 4) == [0, 0, 0, 0, 0]


def test_bench_link_model_arithmetic():
    """bench.py's link_model object (N > 1 lines): the bytes a peer sends (ids at the wire width + u32 counts), the link time at the assumed
    one-way rate and the step DESIGN.md section 5 expects from them -- max(kernels, link) with the gather beside the kernels, their sum
    without."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    lm = bench.link_model(8, [[1.3, 5.12e8]] * 8, 8 * 98_000_000, 8_000_000, 18, 6.5)
    assert lm["peers"] == 7 and lm["bytes_per_peer"] == int(98_000_000 * 18 / 8 + 1_000_000 * 4)
    assert abs(lm["expected_link_ms"] - lm["bytes_per_peer"] / 64e9 * 1e3) < 1e-3
    assert lm["expected_step_ms_overlap"] == max(lm["kernels_ms_max"], lm["expected_link_ms"])
    assert abs(lm["expected_step_ms_sync"] - (lm["kernels_ms_max"] + lm["expected_link_ms"])) < 2e-3
    one = bench.link_model(1, [[1.3, 5.12e8]], 98_000_000, 1_000_000, 32, 1.3)
    assert one["peers"] == 0 and one["expected_link_ms"] == 0.0
