"""CPU, world_size 2, gloo: the multi-GPU path (replicated index, query shards, no per-query
collective).  The search itself needs a GPU, so each rank runs the ORACLE as the stand-in compute;
what is under test is torchpq_amd.distributed (broadcast of the index state, shard bounds, result
order)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import ivfpq_oracle as orc
    from torchpq_amd import distributed as tpd
    fx = load_golden("fx_tiny")
    state = {}
    if rank == 0:
        state = {k[3:]: torch.from_numpy(v.copy()) for k, v in fx.items() if k.startswith("sd.")}
    state = tpd.broadcast_state(state, src=0, device="cpu")
    for k, v in fx.items():
        if k.startswith("sd."):
            assert torch.equal(state[k[3:]], torch.from_numpy(v)), k
    x = torch.from_numpy(fx["queries"])
    k = 10

    def search_fn(xs, kk):
        sd = {n: t.numpy() for n, t in state.items()}
        v, i, _, _, _ = orc.search(xs.numpy(), sd["vq_codec.kmeans.centroids"],
                                   sd["pq_codec.kmeans.centroids"], sd["_storage"], sd["_is_empty"],
                                   sd["_cell_start"], sd["_cell_size"], sd["_address2id"], kk,
                                   int(fx["n_probe"]), use_smart_probing=False)
        return torch.from_numpy(v), torch.from_numpy(i)

    v, i = tpd.sharded_search(search_fn, x, k, gather=True)
    vl, il = tpd.sharded_search(search_fn, x, k, gather=False)
    b, e = tpd.shard_bounds(x.shape[1], rank, world)
    assert torch.equal(v[b:e], vl) and torch.equal(i[b:e], il)
    if rank == 0:
        ev, ei = search_fn(x, k)
        ret["ok"] = bool(torch.equal(v, ev) and torch.equal(i, ei))
    dist.destroy_process_group()


def test_sharded_search_two_ranks_gloo():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get("ok") is True


def test_shard_bounds_cover_and_balance():
    from torchpq_amd.distributed import shard_bounds
    for n in (0, 1, 7, 10000, 10003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[r][1] == b[r + 1][0] for r in range(w - 1))
            sizes = [e - s for s, e in b]
            assert max(sizes) - min(sizes) <= 1


def _groups_worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["RANK"], os.environ["WORLD_SIZE"] = str(rank), str(world)
    from torchpq_amd import distributed as tpd

    # the bulk plane's probe fails on ONE rank only: every rank must learn it and fall back together
    def probe(_group):
        if rank == 1:
            raise RuntimeError("transport down on rank 1")

    def no_rccl_here(*a, **k):  # (this container has no GPU: creating the RCCL group is what fails first)
        raise RuntimeError("no RCCL in this test")

    g = tpd.init_groups("cpu", want_rccl=True, timeout_s=60, probe=probe,
                        create=(lambda: None) if ret["mode"] == "probe" else no_rccl_here)
    assert g.bulk is None and g.bulk_backend == "gloo" and g.bulk_error
    tpd.host_barrier()
    # chunked broadcast: 1000 bytes per call over a 10 KB + a bool + an empty tensor
    state = {}
    if rank == 0:
        gen = torch.Generator().manual_seed(5)
        state = {"a": torch.randint(0, 255, (7, 1501), generator=gen, dtype=torch.uint8),
                 "b": torch.randn(33, 65, generator=gen), "flag": torch.tensor(True),
                 "i": torch.arange(700, dtype=torch.int64), "empty": torch.empty(0, 4)}
    out = tpd.broadcast_state(state, src=0, device="cpu", bulk_group=g.bulk, chunk_bytes=1000)
    gen = torch.Generator().manual_seed(5)
    exp = {"a": torch.randint(0, 255, (7, 1501), generator=gen, dtype=torch.uint8),
           "b": torch.randn(33, 65, generator=gen), "flag": torch.tensor(True),
           "i": torch.arange(700, dtype=torch.int64), "empty": torch.empty(0, 4)}
    ok = all(torch.equal(out[k], exp[k]) and out[k].dtype == exp[k].dtype for k in exp)
    ret[f"ok{rank}"] = bool(ok)
    ret[f"err{rank}"] = g.bulk_error
    dist.destroy_process_group()


def test_rccl_probe_failure_on_one_rank_falls_back_on_all_and_chunked_broadcast():
    for mode in ("probe", "create"):
        mgr = mp.Manager()
        ret = mgr.dict()
        ret["mode"] = mode
        port = 29500 + ((os.getpid() + 7 + len(mode)) % 2000)
        mp.spawn(_groups_worker, args=(2, port, ret), nprocs=2, join=True)
        assert ret.get("ok0") is True and ret.get("ok1") is True
        assert ret["err0"] and ret["err1"]
        if mode == "probe":
            assert "rank 1" in ret["err1"] and "another rank" in ret["err0"]


def test_object_collectives_follow_the_groups_backend(monkeypatch):
    """the metadata broadcasts were forced onto device=cpu, which an NCCL-only default group
    (`init_process_group("nccl")`, the recipe INTEGRATION.md showed) cannot serve.  The device now follows the
    group: CPU where a CPU backend exists, torch's own choice (None) elsewhere."""
    from torchpq_amd import distributed as tpd
    for cfg, cpu in (("cuda:nccl", False), ("nccl", False), ("cpu:gloo,cuda:nccl", True), ("gloo", True)):
        monkeypatch.setattr(dist, "get_backend_config", lambda group=None, c=cfg: c)
        assert tpd.has_cpu_backend(None) is cpu
        assert tpd.object_device(None) == (torch.device("cpu") if cpu else None)
    # the branch torch < 2.1 takes (no get_backend_config): the plain backend name
    monkeypatch.setattr(dist, "get_backend_config", lambda group=None: (_ for _ in ()).throw(AttributeError("x")))
    monkeypatch.setattr(dist, "get_backend", lambda group=None: "nccl")
    assert tpd.object_device(None) is None


def _gloo_worker_control_group(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torchpq_amd import distributed as tpd
    # pretend the default group is NCCL-only: init_groups must then create its own gloo control group and run the
    # MIN over IT; broadcast_state over that group still announces its metadata on the CPU
    real = tpd.has_cpu_backend
    tpd.has_cpu_backend = lambda group=None: False if group is None else real(group)
    try:
        g = tpd.init_groups(device="cpu", want_rccl=True, timeout_s=60, create=lambda: None,
                            probe=lambda grp: None)
    finally:
        tpd.has_cpu_backend = real
    assert g.control is not None and "gloo" in str(dist.get_backend(g.control))
    tpd.host_barrier(g.control)
    st = tpd.broadcast_state({"a": torch.arange(5)} if rank == 0 else {}, src=0, device="cpu", group=g.control)
    if rank == 1:
        ret["ok"] = bool(torch.equal(st["a"], torch.arange(5))) and g.bulk_backend == "nccl"
    dist.destroy_process_group()


def test_init_groups_brings_its_own_control_group_when_the_default_has_no_cpu_backend():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_gloo_worker_control_group, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get("ok") is True
