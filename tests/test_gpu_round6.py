"""GPU, round 6.

* VERDICT r5 #1b: `IVFPQIndex.search()` -- the fused LUT, the scan layout, a batch large enough for the large-batch route
  (four-wave workgroups over the 16-bit table + finish kernel) -- at the FULL configs[1] and configs[3] sizes, a sample of
  the batch bit-compared with the C oracle (the reference's arithmetic, ivfpq_topk.cu:822-971), addresses beyond 2^24
  included;
* the large-batch route of the short codes (m = 8, 16, 32: fp32 table, `scan_finish_exact_kernel` walking
  scan_layout's 32 / 16 / 8-blocks, exact entries from the codebook in LDS or from the caller's table) against the
  reference-layout kernel, bit for bit (reference dispatch: fn/IVFPQTopk.py:39-104);
* the redone-query count on the routes that end with the flag-gated exact kernel.
"""
import os
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _run(**kw):
    from torchpq_amd import kernels as K
    from dump_route_check import run
    return run(K, iters=1, check=True, **kw)


# ---------------------------------------------------------------------------------------------
# the timed routes at full size, against the oracle
# ---------------------------------------------------------------------------------------------
def _route_of_last_search(idx):
    return idx._ivfpq_topk._scan.last_route()


def test_search_at_the_full_c2_shape_equals_the_oracle():
    """configs[1]: 1 M vectors trained / added, n_cells = 1024, m = 64, n_probe = 32, k = 100, 10 000 queries in ONE
    search() call -- the call bench.py times -- and 32 of its rows against the C oracle"""
    import bench
    args = bench.parse_args(["--no-secondary"])
    dev = torch.device(DEV)
    synth = bench.SiftLike(args.d, dev)
    base = synth.sample(args.n_base, seed=1)
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    train = base[:, torch.randperm(args.n_base, generator=g, device=dev)[:args.n_train]].contiguous()
    idx, _, _ = bench.build_index(args, dev, base, train)
    idx.n_probe, idx.use_smart_probing = 32, False
    queries = synth.sample(10000, seed=4321)
    vals, ids = idx.search(queries, k=100)
    torch.cuda.synchronize()
    assert _route_of_last_search(idx) == "dump_sel16"
    oc = bench.oracle_sample_check(idx, queries, vals, ids, 100, n_sample=32)
    assert oc["queries_checked"] == 32 and oc["rows_fully_equal"] == 32, oc
    assert oc["ids_equal_to_oracle"] == 1.0 and oc["values_bit_equal"] is True
    # smart probing on (per-query n_probe): the same route, the same bits
    idx.use_smart_probing = True
    vals, ids = idx.search(queries, k=100)
    oc = bench.oracle_sample_check(idx, queries, vals, ids, 100, n_sample=16)
    assert oc["rows_fully_equal"] == oc["queries_checked"], oc


def test_search_at_the_100m_slot_c4_shape_equals_the_oracle():
    """configs[3] on one GPU: 100 M slots (6.4 GB of codes), n_cells = 16 384, n_probe = 64, k = 100, 10 000 queries in one
    search() call; 32 rows against the C oracle -- most of their addresses lie beyond 2^24, where the reference kernel's
    fp32 address field is no longer exact (ivfpq_topk.cu:13-17)"""
    import bench
    free, _ = torch.cuda.mem_get_info()
    if free < 48 * 2 ** 30:
        pytest.skip("needs ~40 GB of HBM")
    dev = torch.device(DEV)
    idx = bench.fabricate_index(dev, 128, 64, 16384, 100_000_000, seed=1236)
    idx.n_probe, idx.use_smart_probing = 64, False
    g = torch.Generator(device=dev)
    g.manual_seed(4236)
    queries = torch.randn(128, 10000, generator=g, device=dev)
    vals, ids = idx.search(queries, k=100)
    torch.cuda.synchronize()
    assert _route_of_last_search(idx) == "dump_sel16"
    host = {}
    oc = bench.oracle_sample_check(idx, queries, vals, ids, 100, n_sample=32, host=host)
    assert oc["rows_fully_equal"] == 32 and oc["values_bit_equal"] is True and oc["ids_equal_to_oracle"] == 1.0, oc
    assert oc["max_address_checked"] > 2 ** 24 and oc["addresses_beyond_2p24"] > 1000, oc
    # the cold variant of bench.py (every cell probed exactly once): the same kernels through search_cells
    cold = bench.c4_cold_variant(idx, dev, None, 100, 2, host)
    assert cold["oracle_check"]["rows_fully_equal"] == cold["oracle_check"]["queries_checked"] >= 30, cold
    assert cold["roofline"]["reads_of_each_code_byte_per_launch"] <= 1.0


# ---------------------------------------------------------------------------------------------
# short codes on the large-batch route
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,ds", [(32, 4), (32, 1), (16, 8), (16, 2), (8, 16), (8, 4)])
@pytest.mark.parametrize("fused", [True, False])
@pytest.mark.parametrize("nq,k,n_probe,cell", [(1024, 100, 16, 244), (1250, 10, 32, 61), (3000, 248, 8, 977),
                                               (2500, 1, 12, 300)])
def test_short_codes_on_the_large_batch_route(m, ds, fused, nq, k, n_probe, cell):
    from torchpq_amd import kernels as K
    if not fused:   # a caller's table rides the route behind long scans only (kDumpLutMinSlots): make this one long
        n_probe, cell = 32, 977
    route = K.IVFPQTopkHip(m=m).route(nq, k, 1, ds, n_probe, n_probe * cell, has_lut=not fused)
    assert route == "dump_f32", route
    out = _run(m=m, ds=ds, nc=2048 if fused else 256, cell=cell, n_probe=n_probe, k=k, nq=nq, fused=fused, skew=True,
               holes=True)
    assert out["equal"], out


@pytest.mark.parametrize("m,ds,expect", [(32, 8, "one_launch_finish"), (12, 2, "one_launch_finish"),
                                         (24, 4, "one_launch_finish"), (64, 4, "one_launch_finish")])
def test_shapes_the_route_declines_still_equal_the_reference_layout(m, ds, expect):
    """m * ds > 128 (the finish kernel's codebook would not fit the LDS), block structures the route is not built for"""
    from torchpq_amd import kernels as K
    assert K.IVFPQTopkHip(m=m).route(1500, 50, 1, ds, 8, 8 * 244, has_lut=False) == expect
    out = _run(m=m, ds=ds, nc=1024, cell=244, n_probe=8, k=50, nq=1500, fused=True, skew=True, holes=False)
    assert out["equal"], out


def test_index_search_m32_large_batch_equals_small_batches_and_the_oracle():
    """IVFPQIndex.search() at m = 32 (ds = 4: the fused LUT): one batch of 3 000 queries (dump_f32 with a split tail) ==
    the same queries in batches of 500 (the one-launch finish) == the C oracle on a sample"""
    import bench
    from torchpq_amd.index import IVFPQIndex
    rng = np.random.default_rng(12)
    d, n, nq = 128, 150_000, 3000
    base = torch.from_numpy(np.abs(rng.standard_normal((d, n)) * 30).round().astype(np.float32)).to(DEV)
    xq = torch.from_numpy(np.abs(rng.standard_normal((d, nq)) * 30).round().astype(np.float32)).to(DEV)
    np.random.seed(12)
    idx = IVFPQIndex(d_vector=d, n_subvectors=32, n_cells=512, initial_size=64, device=DEV)
    idx.train(base[:, :40000].contiguous())
    idx.add(base)
    idx.remove(ids=torch.arange(0, 3000, 3, device=DEV))
    idx.n_probe = 24
    for smart in (False, True):
        idx.use_smart_probing = smart
        v, i = idx.search(xq, k=100)
        assert _route_of_last_search(idx) == "dump_f32"
        parts = [idx.search(xq[:, b:b + 500].contiguous(), k=100) for b in range(0, nq, 500)]
        assert _route_of_last_search(idx) == "one_launch_finish"
        torch.cuda.synchronize()
        assert torch.equal(v, torch.cat([p[0] for p in parts])) and torch.equal(i, torch.cat([p[1] for p in parts]))
        oc = bench.oracle_sample_check(idx, xq, v, i, 100, n_sample=48)
        assert oc["rows_fully_equal"] == oc["queries_checked"], oc


# ---------------------------------------------------------------------------------------------
# the redone-query count on the routes that end with the exact kernel
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,ds", [(64, 2), (32, 4)])
def test_queries_redone_exactly_are_counted_on_the_large_batch_routes(m, ds):
    """tables that cannot be scaled / bounded flag their query in the scan's prologue; scan_ref_kernel redoes them and
    leaves its mark -- last_redone() used to read 0 here whatever happened (it looked for the one-launch finisher's
    marker in a word that holds the selection band on these routes)"""
    from torchpq_amd import kernels as K
    g = torch.Generator(device=DEV)
    g.manual_seed(7)
    nc, cell, n_probe, k, nq = 512, 200, 8, 50, 1250
    sizes = torch.full((nc,), cell, device=DEV, dtype=torch.long)
    start = torch.cumsum(sizes + 13, 0) - sizes - 13
    n_slots = int((sizes + 13).sum().item())
    storage = torch.randint(0, 256, (m // 4, n_slots, 4), generator=g, device=DEV, dtype=torch.uint8)
    codebook = torch.randn(m, ds, 256, generator=g, device=DEV)
    query = torch.randn(m * ds, nq, generator=g, device=DEV)
    bad = [3, 700, 1100, 1249]
    for q in bad:
        query[:, q] = 1e19        # |q|^2 overflows fp32: every entry -inf
    cells = torch.rand(nq, nc, generator=g, device=DEV).argsort(1)[:, :n_probe].contiguous()
    cs, sz = start[cells].contiguous(), sizes[cells].contiguous()
    npl = torch.full((nq,), n_probe, device=DEV, dtype=torch.long)
    scan = K.IVFPQTopkHip(m=m)
    packed = K.PackCodesHip()(storage)
    scan.keep_workspace = True
    got = scan.topk_fused(storage, query, codebook, None, cs, sz, npl, k, packed=packed, slots_hint=n_probe * cell)
    torch.cuda.synchronize()
    assert scan.last_route() in ("dump_sel16", "dump_f32")
    assert scan.last_redone(nq) == len(bad)
    ref = scan.topk_fused(storage, query, codebook, None, cs, sz, npl, k, packed=None, slots_hint=n_probe * cell)
    torch.cuda.synchronize()
    assert torch.equal(got[1], ref[1])
    # a clean batch counts zero
    query[:, bad] = 0.5
    scan.topk_fused(storage, query, codebook, None, cs, sz, npl, k, packed=packed, slots_hint=n_probe * cell)
    torch.cuda.synchronize()
    assert scan.last_redone(nq) == 0


def test_band_overflow_by_mass_ties_is_redone_and_counted():
    """every slot of the probed cells carries the SAME code: all fast values tie, the band holds thousands of candidates,
    the finish kernel's exact list overflows -> flag -> exact redo; the result still equals the reference-layout kernel
    (value desc, address asc) and the count says how many queries paid for it"""
    from torchpq_amd import kernels as K
    g = torch.Generator(device=DEV)
    g.manual_seed(9)
    m, ds, nc, cell, n_probe, k, nq = 64, 2, 256, 300, 6, 100, 1100
    n_slots = nc * cell
    storage = torch.randint(0, 256, (m // 4, n_slots, 4), generator=g, device=DEV, dtype=torch.uint8)
    tied_cells = torch.arange(0, 40, device=DEV)                 # cells 0 .. 39: one code everywhere
    storage.view(m // 4, nc, cell, 4)[:, tied_cells] = 17
    codebook = torch.randn(m, ds, 256, generator=g, device=DEV)
    query = torch.randn(m * ds, nq, generator=g, device=DEV)
    start = torch.arange(nc, device=DEV) * cell
    sizes = torch.full((nc,), cell, device=DEV, dtype=torch.long)
    cells = torch.rand(nq, nc, generator=g, device=DEV).argsort(1)[:, :n_probe].contiguous()
    cells[:200, :4] = tied_cells[:4]                             # 200 queries probe four all-tied cells and nothing else:
    cs, sz = start[cells].contiguous(), sizes[cells].contiguous()
    npl = torch.full((nq,), n_probe, device=DEV, dtype=torch.long)
    npl[:200] = 4                                                # 1 200 equal values, the best 100 = the 100 lowest addresses
    scan = K.IVFPQTopkHip(m=m)
    packed = K.PackCodesHip()(storage)
    scan.keep_workspace = True
    got = scan.topk_fused(storage, query, codebook, None, cs, sz, npl, k, packed=packed, slots_hint=n_probe * cell)
    torch.cuda.synchronize()
    assert scan.last_route() == "dump_sel16"
    redone = scan.last_redone(nq)
    ref = scan.topk_fused(storage, query, codebook, None, cs, sz, npl, k, packed=None, slots_hint=n_probe * cell)
    torch.cuda.synchronize()
    assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    assert 200 <= redone <= 220, redone
