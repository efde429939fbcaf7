"""The multi-GPU path WITH THE HIP ENGINE as the per-rank encoder (tests/test_parallel_gloo.py covers the same plumbing on
the CPU with the oracle as the encoder): parallel.encode_sharded -- byte-balanced contiguous shards, one variable-length
gather to rank 0 -- and bench.py's self-launching `--gpus N`.

One GPU is what the test box has, so: world = 1 on RCCL (device tensors, HIP pack / unpack kernels, the real process
group), and world = 2 / 3 with the ranks SHARING GPU 0 and gloo carrying the gather.  The peer-to-peer RCCL transfers
themselves need a multi-GPU node (the driver's scaling run)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_world(world, backend, tmp_path):
    out = str(tmp_path / ("result_%s_%d.txt" % (backend, world)))
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "sharded_worker.py"), backend, out], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-1500:] for l in logs)
    assert open(out).read() == "ok", open(out).read()


def test_encode_sharded_hip_engine_world1_rccl(tmp_path):
    _run_world(1, "nccl", tmp_path)


@pytest.mark.parametrize("world", [2, 3])
def test_encode_sharded_hip_engine_ranks_share_gpu_gloo(world, tmp_path):
    _run_world(world, "gloo", tmp_path)


@pytest.mark.parametrize("kind,extra", [("ascii", ["--docs", "30000"]), ("zipf", ["--docs", "6000"])])
def test_bench_self_launches_two_ranks(kind, extra):
    """`python bench.py --gpus 2` with no launcher around it: the parent starts the ranks before touching the GPU; on a
    one-GPU box the ranks share the device and the bench switches the gather to gloo by itself (and says so)."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--kind", kind, "--steps", "3", "--warmup", "1",
                        "--cpu-passes", "0", "--decode-steps", "0", "--host-steps", "0", "--single-docs", "0"] + extra,
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["ids_total"] > 0
    assert len(d["node"]["per_gpu_kernel_ms"]) == 2 and d["node"]["MBps_kernels_only"] >= d["node"]["MBps_gather_inclusive"] * 0.5
    assert d["config"]["baseline_config"] == ("configs[3]" if kind == "ascii" else "configs[4]")
    if kind == "zipf":
        b = d["node"]["per_gpu_input_bytes"]
        assert abs(b[0] - b[1]) <= 2 * 32768 + 64 and "bytes" in d["config"]["sharding"]


def test_node_api_single_device_and_rccl_loopback(tk, test_vocab, monkeypatch):
    """The native multi-GPU entry (tk_node_create / tk_node_encode_batch): with one device the run never touches RCCL; with
    TK_NODE_FORCE_RCCL=1 the same run travels through the whole exchange -- 18-bit pack, ncclSend / ncclRecv to itself
    inside one group, unpack, offsets -- which is everything but a second GPU."""
    import numpy as np
    import corpus
    import helpers
    orc = helpers.oracle_for(test_vocab)
    data, offs = corpus.generate("zipf", 1200, seed=corpus.BASE_SEED + 4)
    eids, eoo = orc.encode_batch(data, offs, True, True, threads=4)
    for force in ("", "1"):
        if force:
            monkeypatch.setenv("TK_NODE_FORCE_RCCL", "1")
        nd = tk.Node(test_vocab["tokens"], test_vocab["num_special"], test_vocab["bos"], test_vocab["eos"], devices=(0,))
        assert nd.n_devices() == 1
        for _ in range(2):                                       # buffers are reused by the second call
            ids, oo = nd.encode_batch(data, offs, True, True)
            assert np.array_equal(ids, eids) and np.array_equal(oo, eoo)
        ids, oo = nd.encode_batch(data[:0], offs[:1], True, True)    # no documents
        assert len(ids) == 0 and oo.tolist() == [0]
        # caller-owned pinned buffers (tk_node_encode_batch_pinned): nothing allocated per call; a buffer that is too small is refused
        h_data, h_offs = tk.host_empty(len(data), np.uint8), tk.host_empty(len(offs), np.uint64)
        h_data[:] = data
        h_offs[:] = offs
        h_ids, h_oo = tk.host_empty(len(data) + 2 * len(offs), np.uint32), tk.host_empty(len(offs), np.uint64)
        for _ in range(2):
            n_ids = nd.encode_batch_into(h_data, h_offs, h_ids, h_oo, True, True)
            assert n_ids == len(eids) and np.array_equal(h_ids[:n_ids], eids) and np.array_equal(h_oo, eoo)
        with pytest.raises(tk.TokenizerError):
            nd.encode_batch_into(h_data, h_offs, h_ids[:100], h_oo, True, True)
        t = nd.last_timing()
        assert t["gather_ms"] >= 0.0
        nd.close()
    with pytest.raises(tk.TokenizerError):
        tk.Node(test_vocab["tokens"], test_vocab["num_special"], test_vocab["bos"], test_vocab["eos"], devices=(0, 0))


def test_node_n_devices_on_one_gpu(tmp_path):
    """csrc/tk_node.cpp's N > 1 branch (it has only ever met a self-loop: this pool hands out one GPU) with N = 2, 3, 8 contexts on
    GPU 0 and the test transport of the development build -- ascii, zipf, a batch whose runs are mostly empty, no documents at all;
    bit-exact against the oracle, byte balance asserted (tests/node_d2d_worker.py)."""
    lib = os.path.join(ROOT, "tekken-rs_amd", "libtekken_hip_ablate.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "tekken-rs_amd"), "ablate"])
    out = str(tmp_path / "node_d2d.txt")
    env = dict(os.environ, TK_HIP_LIB=lib, TK_NODE_TRANSPORT="d2d")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "node_d2d_worker.py"), out], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    assert open(out).read() == "ok", open(out).read()


def test_shipped_library_has_no_test_transport():
    """TK_NODE_TRANSPORT is read by the development build only: the shipped library refuses a device listed twice whatever the
    environment says."""
    import importlib
    tk = importlib.import_module("tekken-rs_amd")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import helpers
    v = helpers.small_trained_vocab()
    os.environ["TK_NODE_TRANSPORT"] = "d2d"
    try:
        with pytest.raises(tk.TokenizerError):
            tk.Node(v["tokens"], v["num_special"], v["bos"], v["eos"], devices=(0, 0))
    finally:
        del os.environ["TK_NODE_TRANSPORT"]
