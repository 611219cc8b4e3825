"""Multi-GPU search: queries shard across ranks, the index is replicated.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" in the CPU
tests).  The only collective is a broadcast of the index buffers at load time (RCCL, probed;
staged over gloo when the probe fails -- init_groups); searching needs no communication -- every query touches read-only state, results stay on the owning
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


# ---- process groups -----------------------------------------------------------------------------
# Two planes.  CONTROL (rendezvous, barriers, metadata, timings): gloo, CPU tensors -- it cannot hang on
# a GPU transport problem.  BULK (the one collective of the path: the index broadcast at load): RCCL over
# xGMI, created eagerly, PROBED with a small all-reduce, and used only when every rank's probe came back;
# otherwise the broadcast is staged through the hosts over gloo and the record says so.  Searching needs
# neither plane.
class Groups:
    def __init__(self, control=None, bulk=None, bulk_backend="gloo", bulk_error=None):
        self.control, self.bulk, self.bulk_backend, self.bulk_error = control, bulk, bulk_backend, bulk_error


def has_cpu_backend(group=None):
    """does `group` (None = the default group) carry a CPU backend (gloo / mpi)?  A group made by
    `init_process_group("nccl")` does not: CPU tensors -- and object collectives forced onto the CPU -- fail on it
    with "No backend type associated with device type cpu"."""
    try:
        cfg = str(dist.get_backend_config(group)).lower()
    except Exception:  # noqa: BLE001 -- older torch: the plain backend name
        cfg = str(dist.get_backend(group)).lower()
    return any(b in cfg for b in ("gloo", "mpi", "ucc"))


def object_device(group=None):
    """the `device=` of the metadata (object) collectives: the CPU wherever the group has a CPU backend -- the control
    plane must not depend on a GPU transport --, None (torch picks the group's own device) on an NCCL-only group"""
    return torch.device("cpu") if has_cpu_backend(group) else None


def host_barrier(group=None):
    """barrier on the control plane (a 1-element CPU all-reduce): callers synchronize their device first.
    (`group`: Groups.control when init_groups had to bring its own gloo group; None = the default group)"""
    if has_cpu_backend(group):
        t = torch.zeros(1, dtype=torch.int32)
        dist.all_reduce(t, group=group)
    else:  # an NCCL-only group handed in by the caller: a device barrier is all there is
        dist.barrier(group=group)


def init_groups(device=None, want_rccl=True, timeout_s=300.0, probe=None, create=None):
    """default process group = gloo (control plane); plus, when `want_rccl`, an RCCL group bound to `device`
    and verified by `probe(group)` (default: an all-reduce of one int32 on the device).  Every rank learns
    whether ALL probes succeeded (a MIN over the control plane) -- a transport that fails on one rank only
    must not leave the others waiting inside a collective.  `create` / `probe` replace the group's
    creation / its check (validation hooks of bench.py and the CPU tests).  Called on an NCCL-only default
    group (`init_process_group("nccl")`), it creates a gloo control group of its own: `Groups.control`."""
    import time
    from datetime import timedelta
    to = timedelta(seconds=timeout_s)
    if not dist.is_initialized():
        dist.init_process_group("gloo", timeout=to)
    g = Groups(control=None)
    if not has_cpu_backend(None):
        # the caller initialised an NCCL-only default group: the control plane gets a gloo group of its
        # own -- the MIN below and every host_barrier(g.control) are CPU collectives
        g.control = dist.new_group(backend="gloo", timeout=to)
    if not want_rccl:
        return g
    err = None
    bulk = None
    try:
        if create is not None:  # (tests: stand-in for the RCCL group)
            bulk = create()
        else:
            bulk = dist.new_group(backend="nccl", timeout=to, device_id=torch.device(device))
        if probe is None:
            # The probe runs on a SIDE stream and is awaited by polling an event against the deadline: a transport
            # that never completes on this rank must not park the host inside synchronize() -- the rank would then
            # miss the MIN below and its peers would wait there.  (A hung collective stays queued on the side
            # stream, which nobody else uses; the group is not used again.  What remains: with the default
            # TORCH_NCCL_ASYNC_ERROR_HANDLING the NCCL watchdog may still abort the process after `timeout_s` --
            # the peers then see the control plane's own timeout, not a hang.)
            side = torch.cuda.Stream(device=device)
            done = torch.cuda.Event()
            t = torch.ones(1, dtype=torch.int32, device=device)
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                dist.all_reduce(t, group=bulk)
                done.record(side)
            deadline = time.monotonic() + timeout_s
            while not done.query():
                if time.monotonic() > deadline:
                    raise RuntimeError(f"RCCL probe all-reduce did not complete within {timeout_s:.0f} s")
                time.sleep(0.002)
            torch.cuda.current_stream(device).wait_stream(side)
            if int(t.item()) != dist.get_world_size():
                raise RuntimeError(f"RCCL probe all-reduce returned {int(t.item())}")
        else:
            probe(bulk)
    except Exception as e:  # noqa: BLE001 -- any transport error: fall back, keep the message
        err = f"{type(e).__name__}: {e}"[:300]
    ok = torch.tensor([0 if err else 1], dtype=torch.int32)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=g.control)
    if int(ok.item()) == 1:
        g.bulk, g.bulk_backend = bulk, "nccl"
    else:
        g.bulk_error = err or "the RCCL probe failed on another rank"
    return g


def _broadcast_chunked(t, src, group, chunk_bytes):
    """broadcast a contiguous tensor in pieces of <= chunk_bytes (flat byte view): bounded staging
    buffers whatever the transport does internally, and a 6.4 GB code array never rides on one call"""
    flat = t.view(-1).view(torch.uint8) if t.dtype != torch.bool else t.view(-1)
    n = flat.numel()
    for b in range(0, n, chunk_bytes):
        dist.broadcast(flat[b:min(n, b + chunk_bytes)], src=src, group=group)


def broadcast_state(state, src=0, device=None, group=None, bulk_group=None, chunk_bytes=1 << 30):
    """Broadcast a flat {name: tensor} dict from `src`.  Shapes differ per index (buffers grow),
    so rank `src` first announces (name, shape, dtype) on `group` (the control plane), then every tensor
    is broadcast -- over `bulk_group` (RCCL) when given, else over `group` -- into a freshly allocated
    buffer on `device`, in chunks of <= chunk_bytes.  Returns the dict on every rank."""
    rank = dist.get_rank(group)
    meta = [None]
    if rank == src:
        meta[0] = [(k, tuple(v.shape), str(v.dtype).replace("torch.", "")) for k, v in state.items()
                   if v is not None]
    dist.broadcast_object_list(meta, src=src, group=group, device=object_device(group))
    data_group = bulk_group if bulk_group is not None else group
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
            if u.numel():
                _broadcast_chunked(u, src, data_group, chunk_bytes)
            t = u.to(torch.bool)
        elif t.numel():
            _broadcast_chunked(t, src, data_group, chunk_bytes)
        out[name] = t
    return out


def replicate_index(index, src=0, group=None, bulk_group=None, chunk_bytes=1 << 30):
    """Make every rank's `index` a replica of rank `src`'s (one broadcast per buffer, <= 1 GiB per call; at the
    100 M configuration ~7.3 GB, per-link bound on xGMI -- tens of ms, paid once at load)."""
    sd = index.state_dict() if dist.get_rank(group) == src else {}
    sd = broadcast_state(sd, src=src, device=index.device, group=group, bulk_group=bulk_group,
                         chunk_bytes=chunk_bytes)
    # bytes that crossed the links, for the record (bench.py `index_broadcast_bytes`)
    index.replicated_bytes = int(sum(v.numel() * v.element_size() for v in sd.values()))
    extra = [None]
    if dist.get_rank(group) == src:
        extra[0] = {"n_probe": index.n_probe, "use_smart_probing": index.use_smart_probing,
                    "smart_probing_temperature": index._smart_probing_temperature}
    dist.broadcast_object_list(extra, src=src, group=group, device=object_device(group))
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
