"""One rank of tests/test_gpu_sharded.py: parallel.encode_sharded with the HIP engine as the per-rank encoder.

  python tests/sharded_worker.py <backend: gloo|nccl> <out_path>       (RANK / WORLD_SIZE / MASTER_* from the environment)

gloo: the ranks may share one GPU (the ids travel through host tensors, 18-bit wire format by the numpy codec);
nccl: device tensors, the HIP pack / unpack kernels (parallel.Ids18Codec) -- one rank per GPU.
Rank 0 compares the gathered ids with the oracle over the WHOLE batch and writes "ok" / a reason to out_path."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tools"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)


def main():
    backend, out_path = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    import corpus
    import helpers
    tk = importlib.import_module("tekken-rs_amd")
    par = importlib.import_module("tekken-rs_amd.parallel")
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev_index = rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    verdict = "ok"
    try:
        v = helpers.small_trained_vocab()
        eng = tk.Engine(v["tokens"], v["num_special"], v["bos"], v["eos"], device=dev_index)
        orc = helpers.oracle_for(v)
        if backend == "nccl":
            codec, device = par.Ids18Codec(tk, eng), torch.device("cuda", dev_index)
        else:
            from test_parallel_gloo import NumpyIds18Codec
            codec, device = NumpyIds18Codec(), None

        def enc(d, o, bos, eos):
            return eng.encode_batch(d, o, bos, eos)

        cases = [("zipf", 1500, 0, 4), ("ascii", 3000, 512, 1), ("mixed", 300, 2048, 2)]
        for kind, n, dl, sd in cases:
            data, offs = corpus.generate(kind, n, dl, seed=corpus.BASE_SEED + sd)
            for cdc in (codec, None):
                ids, oo = par.encode_sharded(enc, data, offs, True, True, dst=0, device=device, codec=cdc)
                if rank == 0:
                    eids, eoo = orc.encode_batch(data, offs, True, True, threads=4)
                    if not (np.array_equal(ids, eids) and np.array_equal(oo, eoo)):
                        verdict = "mismatch on %s (codec %s)" % (kind, type(cdc).__name__)
        # fewer documents than ranks: some shards are empty
        data, offs = corpus.generate("ascii", 1, 64, seed=corpus.BASE_SEED)
        ids, oo = par.encode_sharded(enc, data, offs, False, False, dst=0, device=device, codec=codec)
        if rank == 0:
            eids, eoo = orc.encode_batch(data, offs, False, False)
            if not (np.array_equal(ids, eids) and np.array_equal(oo, eoo)):
                verdict = "mismatch on the single-document batch"
        eng.close()
    except Exception as e:  # noqa: BLE001
        verdict = "rank %d raised %r" % (rank, e)
        raise
    finally:
        if rank == 0 or verdict != "ok":
            with open(out_path if rank == 0 else out_path + ".rank%d" % rank, "w") as f:
                f.write(verdict)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
