"""Multi-GPU search: queries shard across ranks, the index is replicated.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in the CPU
tests).  The only collective is a broadcast of the index buffers at load time; searching
needs no communication -- every query touches read-only state, results stay on the owning
rank (optionally gathered for the caller).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_bounds(n, rank, world_size):
    """Contiguous, balanced [begin, end) of n items for `rank`."""
    base, rem = divmod(n, world_size)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_queries(x, rank=None, world_size=None):
    """x [d, n_query] -> this rank's contiguous column block."""
    rank = dist.get_rank() if rank is None else rank
    world_size = dist.get_world_size() if world_size is None else world_size
    b, e = shard_bounds(x.shape[1], rank, world_size)
    return x[:, b:e].contiguous()


def broadcast_state(state, src=0, device=None, group=None):
    """Broadcast a flat {name: tensor} dict from `src`.  Shapes differ per index (buffers grow),
    so rank `src` first announces (name, shape, dtype), then every tensor is broadcast into a
    freshly allocated buffer on `device`.  Returns the dict on every rank."""
    rank = dist.get_rank(group)
    meta = [None]
    if rank == src:
        meta[0] = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in state.items()
                   if v is not None]
    dist.broadcast_object_list(meta, src=src, group=group)
    out = {}
    for name, shape, dtype in meta[0]:
        dt = getattr(torch, dtype)
        if rank == src:
            t = state[name].to(device) if device is not None else state[name]
            t = t.contiguous()
        else:
            t = torch.empty(shape, dtype=dt, device=device)
        if t.dtype == torch.bool:  # not every backend broadcasts bool
            u = t.to(torch.uint8)
            dist.broadcast(u, src=src, group=group)
            t = u.to(torch.bool)
        elif t.numel():
            dist.broadcast(t, src=src, group=group)
        out[name] = t
    return out


def replicate_index(index, src=0, group=None):
    """Make every rank's `index` a replica of rank `src`'s (one broadcast per buffer; at the 100 M
    configuration ~7.3 GB, per-link bound on xGMI -- tens of ms, paid once at load)."""
    sd = index.state_dict() if dist.get_rank(group) == src else {}
    sd = broadcast_state(sd, src=src, device=index.device, group=group)
    # bytes that crossed the links, for the record (bench.py `index_broadcast_bytes`)
    index.replicated_bytes = int(sum(v.numel() * v.element_size() for v in sd.values()))
    extra = [None]
    if dist.get_rank(group) == src:
        extra[0] = {"n_probe": index.n_probe, "use_smart_probing": index.use_smart_probing,
                    "smart_probing_temperature": index._smart_probing_temperature}
    dist.broadcast_object_list(extra, src=src, group=group)
    if dist.get_rank(group) != src:
        index.load_state_dict(sd)
    index.n_probe = extra[0]["n_probe"]
    index._use_smart_probing = extra[0]["use_smart_probing"]
    index._smart_probing_temperature = extra[0]["smart_probing_temperature"]
    return index


def sharded_search(search_fn, x, k, gather=True, group=None):
    """Run `search_fn(x_shard, k) -> (values, ids)` on this rank's query shard.  With gather=True
    every rank receives the full [n_query, k] result in query order (all_gather of the ragged
    shards); with gather=False only the local shard is returned (no collective at all)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    xs = shard_queries(x, rank, world)
    v, i = search_fn(xs, k)
    if not gather:
        return v, i
    n = x.shape[1]
    sizes = [shard_bounds(n, r, world) for r in range(world)]
    vs = [torch.empty(e - b, k, dtype=v.dtype, device=v.device) for b, e in sizes]
    is_ = [torch.empty(e - b, k, dtype=i.dtype, device=i.device) for b, e in sizes]
    dist.all_gather(vs, v.contiguous(), group=group)
    dist.all_gather(is_, i.contiguous(), group=group)
    return torch.cat(vs, 0), torch.cat(is_, 0)
