"""GPU: the large-batch route of the packed scan at m = 64 (csrc/scan_device.h "dump mode": the scan workgroups stream
over the 16-bit selection table and end with their lists of fast values, scan_finish_exact_kernel evaluates the band's
survivors exactly) against the reference-layout kernel, values and addresses bit for bit
(replaces ivfpq_topk.cu:822-971, kernels/IVFPQTopkCuda.py:81-142).

The batch sizes walk the route's own cases: exactly the chip's workgroup slots, a last round dealt in four parts
(1 100, 1 250), in two (2 500), not split (3 000: the rest fills most of a round), below the route (1 000).
"""
import os
import sys

import pytest
import torch

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def _run(**kw):
    from torchpq_amd import kernels as K
    from dump_route_check import run
    return run(K, iters=1, check=True, **kw)


@pytest.mark.parametrize("nq", [1000, 1024, 1100, 1250, 2500, 3000])
@pytest.mark.parametrize("fused", [True, False])
def test_batch_sizes_around_the_workgroup_slots(nq, fused):
    out = _run(m=64, ds=2, nc=1024, cell=300, n_probe=12, k=100, nq=nq, fused=fused, skew=True, holes=True)
    assert out["equal"], out


@pytest.mark.parametrize("ds,k,n_probe,cell", [(1, 1, 16, 244), (2, 10, 1, 244), (2, 248, 16, 977), (1, 200, 40, 61),
                                               (2, 100, 64, 30), (4, 100, 16, 244)])
def test_k_and_sub_vector_lengths_on_a_split_tail(ds, k, n_probe, cell):
    # (ds = 4: m * ds > 128, the finish kernel cannot hold the query -- the one-launch finish takes it)
    out = _run(m=64, ds=ds, nc=2048, cell=cell, n_probe=n_probe, k=k, nq=1250, fused=True, skew=True, holes=(k != 1))
    assert out["equal"], out


@pytest.mark.parametrize("k,cell,n_probe,nq", [(249, 977, 32, 2048), (300, 977, 32, 1500), (500, 977, 16, 1024),
                                               (504, 500, 40, 1200), (400, 244, 32, 1100)])
def test_k_up_to_504_rides_the_route_too(k, cell, n_probe, nq):
    # (the last case -- short cells, no "spread" -- asks for lists of 8 registers: left to the sorted lists)
    out = _run(m=64, ds=2, nc=1024, cell=cell, n_probe=n_probe, k=k, nq=nq, fused=True, skew=True, holes=True)
    assert out["equal"], out


def test_tables_the_16_bit_scale_cannot_hold_go_to_the_exact_kernel():
    """queries whose table has an Inf / a huge entry are flagged by the scan and redone by the exact kernel"""
    from torchpq_amd import kernels as K
    dev = "cuda:0"
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    m, ds, nc, cell, n_probe, k, nq = 64, 2, 512, 200, 8, 50, 1250
    sizes = torch.full((nc,), cell, device=dev, dtype=torch.long)
    start = torch.cumsum(sizes + 13, 0) - sizes - 13
    n_slots = int((sizes + 13).sum().item())
    storage = torch.randint(0, 256, (m // 4, n_slots, 4), generator=g, device=dev, dtype=torch.uint8)
    codebook = torch.randn(m, ds, 256, generator=g, device=dev)
    query = torch.randn(m * ds, nq, generator=g, device=dev)
    query[:, 3] = 1e19        # |q|^2 overflows fp32: every entry -inf
    query[5, 1100] = 3e18     # one component: entries of one sub-quantizer beyond the scale's range
    query[:, 1249] = 0.0
    cells = torch.rand(nq, nc, generator=g, device=dev).argsort(1)[:, :n_probe].contiguous()
    cs, sz = start[cells].contiguous(), sizes[cells].contiguous()
    npl = torch.full((nq,), n_probe, device=dev, dtype=torch.long)
    scan = K.IVFPQTopkHip(m=m)
    packed = K.PackCodesHip()(storage)
    got = scan.topk_fused(storage, query, codebook, None, cs, sz, npl, k, packed=packed, slots_hint=n_probe * cell)
    ref = scan.topk_fused(storage, query, codebook, None, cs, sz, npl, k, packed=None, slots_hint=n_probe * cell)
    torch.cuda.synchronize()
    ok = ~torch.isnan(ref[0]).any(1)   # (NaN values compare unequal to themselves: checked through the addresses)
    assert torch.equal(got[0][ok], ref[0][ok])
    assert torch.equal(got[1], ref[1])


@pytest.mark.parametrize("k,n_probe,nq", [(100, 200, 1300), (10, 70, 2100), (300, 96, 1024), (40, 5, 4000)])
def test_ragged_probe_lists_empty_and_repeated_cells(k, n_probe, nq):
    """per-query n_probe (smart probing), more than 64 probes (the probe table's second round), empty cells, a cell
    listed twice in a row (skipped: ivfpq_topk.cu:864-866), tombstones -- against the reference-layout kernel"""
    from torchpq_amd import kernels as K
    dev = "cuda:0"
    g = torch.Generator(device=dev)
    g.manual_seed(k * 7 + n_probe)
    m, ds, nc = 64, 2, 700
    sizes = (torch.rand(nc, generator=g, device=dev) ** 3 * 900).long()
    sizes[torch.rand(nc, generator=g, device=dev) < 0.15] = 0                      # empty cells
    cap = sizes + 9
    start = torch.cumsum(cap, 0) - cap
    n_slots = int(cap.sum().item())
    storage = torch.randint(0, 256, (m // 4, n_slots, 4), generator=g, device=dev, dtype=torch.uint8)
    is_empty = (torch.rand(n_slots, generator=g, device=dev) < 0.05).to(torch.uint8)
    codebook = torch.randn(m, ds, 256, generator=g, device=dev) * 3
    query = torch.randn(m * ds, nq, generator=g, device=dev) * 3
    cells = torch.rand(nq, nc, generator=g, device=dev).argsort(1)[:, :n_probe].contiguous()
    rep = torch.rand(nq, generator=g, device=dev) < 0.3                            # probe 1 repeats probe 0
    if n_probe > 1:
        cells[rep, 1] = cells[rep, 0]
    cs, sz = start[cells].contiguous(), sizes[cells].contiguous()
    npl = torch.randint(1, n_probe + 1, (nq,), generator=g, device=dev)
    npl[::17] = 0                                                                  # queries that probe nothing
    scan = K.IVFPQTopkHip(m=m)
    packed = K.PackCodesHip()(storage)
    hint = int(sizes.float().mean().item() * n_probe)
    got = scan.topk_fused(storage, query, codebook, is_empty, cs, sz, npl, k, packed=packed, slots_hint=hint)
    ref = scan.topk_fused(storage, query, codebook, is_empty, cs, sz, npl, k, packed=None, slots_hint=hint)
    torch.cuda.synchronize()
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    assert bool(torch.isinf(ref[0][::17]).all())                                   # (-inf, -1) padding rows


def test_index_search_large_batch_equals_small_batches():
    """IVFPQIndex.search(): one batch of 3 000 queries (the large-batch route with its split tail) == the same queries
    in batches of 500 (the one-launch finish), values and ids bit for bit; smart probing on; some items removed"""
    import numpy as np
    from torchpq_amd.index import IVFPQIndex
    dev = "cuda:0"
    rng = np.random.default_rng(11)
    d, n, nq = 128, 120_000, 3000
    base = torch.from_numpy(np.abs(rng.standard_normal((d, n)) * 30).astype(np.float32)).to(dev)
    xq = torch.from_numpy(np.abs(rng.standard_normal((d, nq)) * 30).astype(np.float32)).to(dev)
    np.random.seed(11)
    idx = IVFPQIndex(d_vector=d, n_subvectors=64, n_cells=256, initial_size=64, device=dev)
    idx.train(base[:, :40000].contiguous())
    idx.add(base)
    idx.remove(ids=torch.arange(0, 3000, 3, device=dev))
    idx.n_probe = 24
    for smart in (False, True):
        idx.use_smart_probing = smart
        v, i = idx.search(xq, k=150)
        parts = [idx.search(xq[:, b:b + 500].contiguous(), k=150) for b in range(0, nq, 500)]
        torch.cuda.synchronize()
        assert torch.equal(v, torch.cat([p[0] for p in parts])) and torch.equal(i, torch.cat([p[1] for p in parts]))
