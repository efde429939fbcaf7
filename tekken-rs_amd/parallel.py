"""Multi-GPU driver: one process per GPU, documents sharded whole, ONE variable-length gather.

Documents are independent (reference `encode` is a pure function of one &str,
src/tekkenizer.rs:378-405), so the path shards with no data-path collective; the only exchange
is the gather of the token-id buffers to rank 0 at the end (BASELINE.json north_star, SURVEY
section 8e).  RCCL has no gatherv, so: one all_gather of the per-rank id counts (8 bytes each),
then direct point-to-point transfers -- every peer sends its buffer straight to the root, the
root posts one receive per peer, all inside one batch (ncclGroupStart/End under
`batch_isend_irecv`).  On an MI355X node every peer->root transfer rides its own xGMI link, so
the 7 transfers run concurrently; a ring would be bound by one link.

What bounds N > 1 is that link, so the gather is built around it:
  * a `codec` shrinks the ids on the wire (`Ids18Codec`: 2.25 bytes per id instead of 4, packed and unpacked by HIP
    kernels through the C ABI; ids of every Tekken vocabulary are below 2^18);
  * `wait=False` returns a `PendingGather` instead of blocking: the transfer (and, on the root, the unpacking) of
    batch k then runs beside the kernels of batch k + 1 on a side stream.

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


class Ids18Codec:
    """18-bit wire format of ids, packed / unpacked on the device by the HIP kernels behind the C ABI
    (tk_pack_ids18_device / tk_unpack_ids18_device, include/tekken_hip.h)."""

    def __init__(self, tk, engine):
        self.tk, self.eng = tk, engine

    def packed_numel(self, n_ids):            # int32 elements of the wire buffer
        return (self.tk.ids18_bytes(n_ids) + 3) // 4

    def pack(self, ids):
        import torch
        out = torch.empty(self.packed_numel(ids.numel()), dtype=torch.int32, device=ids.device)
        self.eng.pack_ids18_device(ids.data_ptr(), ids.numel(), out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        return out

    def unpack(self, packed, n_ids, out):
        import torch
        self.eng.unpack_ids18_device(packed.data_ptr(), n_ids, out.data_ptr(), torch.cuda.current_stream().cuda_stream)


class PendingGather:
    """A gather whose transfers are in flight.  result() waits and returns what gather_ids returns."""

    def __init__(self, finish):
        self._finish = finish
        self._res = None

    def result(self):
        if self._finish is not None:
            self._res = self._finish()
            self._finish = None
        return self._res


def gather_ids(local_ids, local_doc_counts, dst=0, group=None, codec=None, wait=True):
    """Variable-length gather of token ids (+ per-document id counts) to rank `dst`.

    local_ids: 1-D int32 tensor (uint32 ids reinterpreted), on the device of the backend.
    local_doc_counts: 1-D integer tensor, ids per local document (travels as int32: a document has fewer than 2^31 ids, and the
    counts of a million documents are 4 MB a peer instead of 8).
    codec: None (ids travel as they are) or an object with packed_numel / pack / unpack (Ids18Codec).
    Returns on dst: (ids tensor, doc_offsets tensor int64[D_total+1]) in rank order = document
    order; on other ranks: (None, None).  wait=False: a PendingGather (call .result() later); on a CUDA device
    the transfers and the unpacking are then ordered on a side stream, so that work enqueued on the current
    stream afterwards runs beside them."""
    import torch
    import torch.distributed as dist
    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    dev = local_ids.device
    on_gpu = dev.type == "cuda"
    local_doc_counts = local_doc_counts.to(torch.int32)
    sizes = torch.tensor([local_ids.numel(), local_doc_counts.numel()], dtype=torch.int64, device=dev)
    all_sizes = [torch.zeros(2, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = [s.tolist() for s in all_sizes]
    side = main = None
    if on_gpu and not wait:
        main = torch.cuda.current_stream(dev)
        side = _side_stream(dev)
        # The caller's tensors are consumed on the side stream: tell the caching allocator, or a block freed on the main
        # stream right after this call (bench.py passes a temporary clone) could be handed out again while the side-stream
        # copy / send is still pending.
        local_ids.record_stream(side)
        local_doc_counts.record_stream(side)

    def on_side(after_main=True):
        if side is None:
            return _null()
        if after_main:
            side.wait_stream(main)             # what the current stream produced so far (the ids) is ready
        return torch.cuda.stream(side)

    if rank == dst:
        # one output buffer, every peer's transfer lands in its slice (no concatenation pass over the gathered ids)
        n_ids = [s[0] for s in all_sizes]
        n_cnt = [s[1] for s in all_sizes]
        with on_side():
            ids = torch.empty(sum(n_ids), dtype=local_ids.dtype, device=dev)
            counts = torch.empty(sum(n_cnt), dtype=torch.int32, device=dev)
            ops, wires = [], []
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
                    if codec is None:
                        ops.append(dist.P2POp(dist.irecv, ids_r, peer, group))
                    else:
                        w = torch.empty(codec.packed_numel(n_ids[r]), dtype=torch.int32, device=dev)
                        wires.append((w, n_ids[r], ids_r))
                        ops.append(dist.P2POp(dist.irecv, w, peer, group))
                if n_cnt[r]:
                    ops.append(dist.P2POp(dist.irecv, cnt_r, peer, group))
            works = dist.batch_isend_irecv(ops) if ops else []

        def finish_root():
            with on_side(False):
                for w in works:
                    w.wait()
                for w, n, out in wires:
                    codec.unpack(w, n, out)
                offs = torch.zeros(counts.numel() + 1, dtype=torch.int64, device=dev)
                torch.cumsum(counts, 0, dtype=torch.int64, out=offs[1:])
            if side is not None:
                side.synchronize()
                # allocated under the side stream, used (and eventually freed) by the caller on the main stream
                ids.record_stream(main)
                offs.record_stream(main)
            return ids, offs

        return finish_root() if wait else PendingGather(finish_root)

    with on_side():
        ops, keep = [], []
        peer = dist.get_global_rank(group, dst) if group is not None else dst
        if local_ids.numel():
            wire = local_ids if codec is None else codec.pack(local_ids)
            keep.append(wire)
            ops.append(dist.P2POp(dist.isend, wire, peer, group))
        if local_doc_counts.numel():
            ops.append(dist.P2POp(dist.isend, local_doc_counts, peer, group))
        works = dist.batch_isend_irecv(ops) if ops else []

    def finish_peer():
        with on_side(False):
            for w in works:
                w.wait()
        if side is not None:
            side.synchronize()
        keep.clear()
        return None, None

    return finish_peer() if wait else PendingGather(finish_peer)


def _null():
    import contextlib
    return contextlib.nullcontext()


_SIDE = {}


def _side_stream(dev):
    import torch
    key = str(dev)
    if key not in _SIDE:
        _SIDE[key] = torch.cuda.Stream(device=dev)
    return _SIDE[key]


def encode_sharded(encode_fn, data, offs, add_bos=True, add_eos=True, dst=0, group=None, device=None, codec=None):
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
    g_ids, g_offs = gather_ids(t_ids, t_cnt, dst=dst, group=group, codec=codec)
    if rank != dst:
        return None, None
    return g_ids.cpu().numpy().view(np.uint32), g_offs.cpu().numpy().astype(np.uint64)
