"""Multi-GPU driver: one process per GPU, documents sharded whole, ONE variable-length gather.

Documents are independent (reference `encode` is a pure function of one &str,
src/tekkenizer.rs:378-405), so the path shards with no data-path collective; the only exchange
is the gather of the token-id buffers to rank 0 at the end (BASELINE.json north_star, SURVEY
section 8e).  RCCL has no gatherv, so: one all_gather of the per-rank id counts (8 bytes each),
then direct point-to-point transfers -- every peer sends its buffer straight to the root, the
root posts one receive per peer, all inside one batch (ncclGroupStart/End under
`batch_isend_irecv`).  On an MI355X node every peer->root transfer rides its own xGMI link, so
the 7 transfers run concurrently; a ring would be bound by one link.

`torch.distributed` is plumbing: backend "nccl" (= RCCL) for device buffers, "gloo" for the CPU
tests of this logic.
"""
import numpy as np


def shard_by_bytes(offs, world_size):
    """Contiguous document ranges with balanced BYTES (not counts): returns world_size+1 cut points.

    offs: uint64[D+1] document offsets.  Rank r owns documents [cuts[r], cuts[r+1])."""
    offs = np.asarray(offs, dtype=np.uint64)
    n_docs = len(offs) - 1
    total = int(offs[-1]) - int(offs[0])
    cuts = [0]
    for r in range(1, world_size):
        target = int(offs[0]) + total * r // world_size
        c = int(np.searchsorted(offs, np.uint64(target), side="left"))
        c = min(max(c, cuts[-1]), n_docs)
        cuts.append(c)
    cuts.append(n_docs)
    return cuts


def gather_ids(local_ids, local_doc_counts, dst=0, group=None):
    """Variable-length gather of token ids (+ per-document id counts) to rank `dst`.

    local_ids: 1-D int32 tensor (uint32 ids reinterpreted), on the device of the backend.
    local_doc_counts: 1-D int64 tensor, ids per local document.
    Returns on dst: (ids tensor, doc_offsets tensor int64[D_total+1]) in rank order = document
    order; on other ranks: (None, None)."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    dev = local_ids.device
    sizes = torch.tensor([local_ids.numel(), local_doc_counts.numel()], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = [s.tolist() for s in all_sizes]
    if rank == dst:
        # one output buffer, every peer's transfer lands in its slice (no concatenation pass over the gathered ids)
        n_ids = [s[0] for s in all_sizes]
        n_cnt = [s[1] for s in all_sizes]
        ids = torch.empty(sum(n_ids), dtype=local_ids.dtype, device=dev)
        counts = torch.empty(sum(n_cnt), dtype=torch.int64, device=dev)
        ops = []
        i0 = c0 = 0
        for r in range(world):
            ids_r, cnt_r = ids[i0:i0 + n_ids[r]], counts[c0:c0 + n_cnt[r]]
            i0 += n_ids[r]
            c0 += n_cnt[r]
            if r == dst:
                ids_r.copy_(local_ids)
                cnt_r.copy_(local_doc_counts)
                continue
            peer = dist.get_global_rank(group, r) if group is not None else r
            if n_ids[r]:
                ops.append(dist.P2POp(dist.irecv, ids_r, peer, group))
            if n_cnt[r]:
                ops.append(dist.P2POp(dist.irecv, cnt_r, peer, group))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        offs = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=dev)
        torch.cumsum(counts, 0, out=offs[1:])
        return ids, offs
    ops = []
    peer = dist.get_global_rank(group, dst) if group is not None else dst
    if local_ids.numel():
        ops.append(dist.P2POp(dist.isend, local_ids, peer, group))
    if local_doc_counts.numel():
        ops.append(dist.P2POp(dist.isend, local_doc_counts, peer, group))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return None, None


def encode_sharded(encode_fn, data, offs, add_bos=True, add_eos=True, dst=0, group=None, device=None):
    """Shard `data/offs` (host numpy, identical on every rank) by bytes, encode the local shard with
    `encode_fn(local_data, local_offs, add_bos, add_eos) -> (ids uint32[T], out_offs uint64[D+1])`
    and gather to `dst`.  Returns (ids uint32 numpy, offsets uint64 numpy) on dst, (None, None) elsewhere."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    offs = np.asarray(offs, dtype=np.uint64)
    cuts = shard_by_bytes(offs, world)
    d0, d1 = cuts[rank], cuts[rank + 1]
    lo, hi = int(offs[d0]), int(offs[d1])
    local_offs = (offs[d0:d1 + 1] - offs[d0]).astype(np.uint64)
    ids, oo = encode_fn(np.asarray(data[lo:hi]), local_offs, add_bos, add_eos)
    dev = device if device is not None else torch.device("cpu")
    t_ids = torch.from_numpy(np.ascontiguousarray(ids).view(np.int32)).to(dev)
    t_cnt = torch.from_numpy(np.diff(oo.astype(np.int64))).to(dev)
    g_ids, g_offs = gather_ids(t_ids, t_cnt, dst=dst, group=group)
    if rank != dst:
        return None, None
    return g_ids.cpu().numpy().view(np.uint32), g_offs.cpu().numpy().astype(np.uint64)
