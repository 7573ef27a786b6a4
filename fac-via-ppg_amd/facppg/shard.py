"""Utterance sharding for offline corpus synthesis on one 8-GPU node (BASELINE config 4).

Utterances are independent (no cross-utterance state anywhere on the path), so the corpus is
partitioned by length-sorted round-robin dealing (balances the sum of lengths per rank), every
rank synthesises its shard with its own copy of the weights, and the only exchanges are
  * one broadcast of the packed weights at start (or each rank reads the checkpoint), and
  * one all_gather of lengths + one padded gather of the audio to rank 0 at the end
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
    {global id: tensor (on CPU)}, elsewhere None.  Two collectives: all_gather of the
    (id, length) table, then a gather of one padded [n_max, len_max] block per rank."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = items[0].device if items else torch.device("cpu")
    if dist.get_backend() == "nccl" and dev.type != "cuda":
        dev = torch.device("cuda", torch.cuda.current_device())
    n_local = torch.tensor([len(items), max([int(t.numel()) for t in items], default=0)], device=dev, dtype=torch.int64)
    table = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(table, n_local)
    n_max = int(max(t[0] for t in table))
    l_max = int(max(t[1] for t in table))
    meta = torch.full((n_max, 2), -1, device=dev, dtype=torch.int64)
    block = torch.zeros(n_max, max(l_max, 1), device=dev, dtype=torch.float32)
    for j, (t, gid) in enumerate(zip(items, indices)):
        meta[j, 0], meta[j, 1] = int(gid), int(t.numel())
        block[j, :t.numel()] = t.to(dev).float()
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    blocks = [torch.empty_like(block) for _ in range(world)] if rank == dst else None
    dist.gather(block, blocks, dst=dst)
    if rank != dst:
        return None
    out = {}
    for m, blk in zip(metas, blocks):
        m, blk = m.cpu(), blk.cpu()
        for j in range(m.shape[0]):
            gid, n = int(m[j, 0]), int(m[j, 1])
            if gid >= 0:
                out[gid] = blk[j, :n].clone()
    return out
