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
