"""Utterance sharding for offline corpus synthesis on one 8-GPU node (BASELINE config 4).

Utterances are independent (no cross-utterance state anywhere on the path), so the corpus is
partitioned by length-sorted round-robin dealing (balances the sum of lengths per rank), every
rank synthesises its shard with its own copy of the weights, and the only exchanges are
  * one broadcast of the packed weights at start (or each rank reads the checkpoint), and
  * all_gathers of the (id, length) tables + ONE gather of each rank's packed audio to rank 0 at the end
over torch.distributed (backend "nccl" = RCCL over xGMI on the GPUs; "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def partition(lengths, world_size):
    """Deal utterance indices, longest first, round-robin to ranks.  Returns list (per rank) of
    index lists; every index appears exactly once."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return [order[r::world_size] for r in range(world_size)]


def batches(indices, lengths, batch_size):
    """Split one rank's (already length-sorted) indices into batches of similar length."""
    return [indices[i:i + batch_size] for i in range(0, len(indices), batch_size)]


def broadcast_state(tensors, src=0):
    """One flat broadcast of a list of tensors (the reference broadcasts 938 tensors one by one,
    distributed.py:100-103)."""
    flat = torch.cat([t.reshape(-1).float() for t in tensors])
    dist.broadcast(flat, src)
    out, off = [], 0
    for t in tensors:
        n = t.numel()
        out.append(flat[off:off + n].view_as(t).to(t.dtype))
        off += n
    return out


def gather_ragged(items, indices, dst=0):
    """Gather variable-length 1-D float tensors from all ranks to ``dst``.

    items: this rank's tensors, indices: their global utterance ids.  Returns on ``dst`` a dict
    {global id: tensor (on CPU)}, elsewhere None.

    Every rank packs its waveforms ONCE into one flat buffer (one ``cat``) next to an (id, length) table; three
    collectives move them: an all_gather of (item count, sample count), an all_gather of the tables padded to the largest
    item count (a few KB), and the flat buffers themselves, point to point at their EXACT sizes (batch_isend_irecv: every
    rank one send, dst all its receives in one group) -- instead of the former [n_max, l_max] block per rank whose rows
    were each padded to the longest utterance (~40 % of the bytes for lengths of 100..400 frames) and filled row by row
    from a Python loop."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = items[0].device if items else torch.device("cpu")
    if dist.get_backend() == "nccl" and dev.type != "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    lens = [int(t.numel()) for t in items]
    counts = torch.tensor([len(items), sum(lens)], device=dev, dtype=torch.int64)
    table = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(table, counts)
    table = torch.stack(table).cpu()
    n_max = int(table[:, 0].max())
    meta = torch.full((max(n_max, 1), 2), -1, dtype=torch.int64)
    if items:
        meta[:len(items), 0] = torch.tensor([int(g) for g in indices], dtype=torch.int64)
        meta[:len(items), 1] = torch.tensor(lens, dtype=torch.int64)
    meta = meta.to(dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    # torch.distributed's gather wants equal sizes on every backend, so the flat buffers go point to point: every rank
    # sends its exact sum of samples to dst, which posts all its receives at once (RCCL: one grouped launch over xGMI)
    total = sum(lens)
    flat = torch.cat([t.reshape(-1).to(device=dev, dtype=torch.float32) for t in items]) if total else torch.zeros(0, device=dev)   # packed once
    flats = None
    if rank == dst:
        flats = [flat if r == dst else torch.empty(int(table[r, 1]), device=dev, dtype=torch.float32) for r in range(world)]
        ops = [dist.P2POp(dist.irecv, flats[r], r) for r in range(world) if r != dst and flats[r].numel()]
    else:
        ops = [dist.P2POp(dist.isend, flat, dst)] if total else []
    if ops:
        for work in dist.batch_isend_irecv(ops):
            work.wait()
    if rank != dst:
        return None
    out = {}
    for m, buf in zip(metas, flats):
        m, buf = m.cpu(), buf.cpu()                          # ONE device->host copy per rank
        off = 0
        for j in range(m.shape[0]):
            gid, n = int(m[j, 0]), int(m[j, 1])
            if gid >= 0:
                out[gid] = buf[off:off + n].clone()
                off += n
    return out
