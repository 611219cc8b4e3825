"""GPU: the scan at BASELINE.json's full sizes, through size-independent properties
(sortedness, layout invariance, split invariance) plus a sampled oracle check."""
import numpy as np
import pytest
import torch

from oracle import c_oracle

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _synthetic(n_cells, cell, m, nq, n_probe, seed, slack=47):
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    sizes = torch.randint(max(1, cell // 2), cell + cell // 2, (n_cells,), generator=g, device=DEV)
    cap = sizes + slack
    start = torch.cumsum(cap, 0) - cap
    n_slots = int(cap.sum().item())
    storage = torch.randint(0, 256, (m // 4, n_slots, 4), generator=g, device=DEV, dtype=torch.uint8)
    lut = torch.randn(m, nq, 256, generator=g, device=DEV) * 50 - 300
    cells = torch.rand(nq, n_cells, generator=g, device=DEV).argsort(1)[:, :n_probe].contiguous() \
        if n_cells <= 4096 else torch.randint(0, n_cells, (nq, n_probe), generator=g, device=DEV)
    return storage, lut, start[cells].contiguous(), sizes[cells].contiguous()


@pytest.mark.parametrize("name,n_cells,cell,m,nq,n_probe,k", [
    ("C2", 1024, 977, 64, 2000, 32, 100),     # SIFT1M shape
    ("C3", 1024, 977, 120, 500, 64, 100),     # GIST1M shape (120 KiB LUT, 16-wave workgroups)
    ("C1", 256, 390, 16, 1000, 8, 10),        # the plumbing config's shape
])
def test_full_size_properties(name, n_cells, cell, m, nq, n_probe, k):
    import torchpq_amd.kernels as K
    storage, lut, cs, sz = _synthetic(n_cells, cell, m, nq, n_probe, seed=hash(name) % 1000)
    npl = torch.full((nq,), n_probe, device=DEV, dtype=torch.long)
    scan = K.IVFPQTopkHip(m=m)
    packed = K.PackCodesHip()(storage)
    v_ref, a_ref = scan.topk(storage, lut, None, cs, sz, npl, n_candidates=k, n_split=1)
    v_pk, a_pk = scan.topk(storage, lut, None, cs, sz, npl, n_candidates=k, packed=packed, n_split=1)
    v_sp, a_sp = scan.topk(storage, lut, None, cs, sz, npl, n_candidates=k, packed=packed, n_split=3)
    assert torch.equal(v_ref, v_pk) and torch.equal(a_ref, a_pk)          # layout invariance
    assert torch.equal(v_ref, v_sp) and torch.equal(a_ref, a_sp)          # split invariance
    assert bool((v_ref[:, 1:] <= v_ref[:, :-1]).all())                    # sorted descending
    # every returned address lies inside one of the query's probed cells, no duplicates
    lo, hi = cs[:, None, :], (cs + sz)[:, None, :]
    inside = ((a_ref[:, :, None] >= lo) & (a_ref[:, :, None] < hi)).any(-1)
    assert bool(inside.all())
    srt = a_ref.sort(1).values
    assert bool((srt[:, 1:] != srt[:, :-1]).all())
    # sampled oracle check (the C restatement on 24 queries)
    sel = torch.arange(0, nq, max(1, nq // 24), device=DEV)[:24]
    ev, ea = c_oracle.scan_topk(storage.cpu().numpy(), lut[:, sel].contiguous().cpu().numpy(), None,
                                cs[sel].cpu().numpy(), sz[sel].cpu().numpy(),
                                npl[sel].cpu().numpy(), k)
    assert np.array_equal(v_ref[sel].cpu().numpy(), ev)
    assert np.array_equal(a_ref[sel].cpu().numpy(), ea)


def test_c4_shape_100m_slots():
    """configs[3]: 100 M vectors, n_cells=16384, m=64, n_probe=64 (6.4 GB of codes -- past the
    Infinity Cache): packed == reference layout, sampled oracle check."""
    import torchpq_amd.kernels as K
    free, _ = torch.cuda.mem_get_info()
    if free < 40 * 2 ** 30:
        pytest.skip("needs ~20 GB of HBM")
    nq, k, n_probe, m = 256, 100, 64, 64
    storage, lut, cs, sz = _synthetic(16384, 6103, m, nq, n_probe, seed=4, slack=9)
    assert storage.shape[1] > 95_000_000
    npl = torch.full((nq,), n_probe, device=DEV, dtype=torch.long)
    scan = K.IVFPQTopkHip(m=m)
    packed = K.PackCodesHip()(storage)
    v_pk, a_pk = scan.topk(storage, lut, None, cs, sz, npl, n_candidates=k, packed=packed)
    v_ref, a_ref = scan.topk(storage, lut, None, cs, sz, npl, n_candidates=k)
    assert torch.equal(v_ref, v_pk) and torch.equal(a_ref, a_pk)
    assert int(a_ref.max()) > 2 ** 24  # addresses beyond the reference kernel's fp32-exact range
    del packed
    sel = torch.tensor([0, 17, 101, 255], device=DEV)
    ev, ea = c_oracle.scan_topk(storage.cpu().numpy(), lut[:, sel].contiguous().cpu().numpy(), None,
                                cs[sel].cpu().numpy(), sz[sel].cpu().numpy(), npl[sel].cpu().numpy(), k)
    assert np.array_equal(v_ref[sel].cpu().numpy(), ev)
    assert np.array_equal(a_ref[sel].cpu().numpy(), ea)


def test_c5_kmeans_full_size_properties():
    """configs[4] (MultiKMeans, 64 sub-problems x 64 dims x 1 M points, 256 clusters: 16.4 GB):
    identity on the centroids, sampled bit-exact oracle check, linearity of the update
    (sum_c count_c * centroid_c == sum_i x_i), Lloyd monotonicity."""
    import torchpq_amd.kernels as K
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * 2 ** 30:
        pytest.skip("needs ~35 GB of HBM")
    l, d, n, k = 64, 64, 1_000_000, 256
    g = torch.Generator(device=DEV)
    g.manual_seed(5)
    data = torch.randn(l, d, n, generator=g, device=DEV)
    cent = data[:, :, torch.randperm(n, generator=g, device=DEV)[:k]].contiguous()
    ms = K.MaxSimHip(distance="euclidean")
    # the centroids are their own nearest centroid, at similarity exactly 0 (2s - s - s)
    v_c, i_c = ms(cent, cent, dim=2)
    assert torch.equal(i_c, torch.arange(k, device=DEV).expand(l, k)) and bool((v_c == 0).all())
    vals, labels = ms(data, cent, dim=2)
    assert labels.dtype == torch.int64 and int(labels.min()) >= 0 and int(labels.max()) < k
    assert bool((vals <= 0).all())
    # sampled oracle check: 4 sub-problems x 64 points, bit-exact (fma chains)
    for b in (0, 21, 42, 63):
        pts = torch.arange(b * 997, n, n // 64, device=DEV)[:64]
        ev, ei = c_oracle.max_sim(data[b:b + 1][:, :, pts].cpu().numpy(), cent[b:b + 1].cpu().numpy(),
                                  "euclidean", "expanded")
        assert np.array_equal(vals[b, pts].cpu().numpy(), ev[0])
        assert np.array_equal(labels[b, pts].cpu().numpy(), ei[0])
    new_cent = K.ComputeCentroidsHip()(data, labels, k)
    counts = torch.zeros(l, k, device=DEV).scatter_add_(1, labels, torch.ones(l, n, device=DEV))
    assert bool((counts > 0).all())
    total = data.sum(-1)                                        # [l, d]
    recon = (new_cent * counts[:, None, :]).sum(-1)             # sum_c count_c * centroid_c
    scale = float(data.abs().sum(-1).max())
    assert float((total - recon).abs().max()) <= 2e-4 * scale
    vals2, _ = ms(data, new_cent, dim=2)
    assert float(vals2.double().sum()) > float(vals.double().sum())   # inertia went down


def test_c2_index_full_size_round_trip():
    """configs[1] shape through the index: train on 100 k, add 1 M, query with stored vectors:
    a vector finds itself (encode -> scan -> id), its value is -|x - decode(encode(x))|^2, and
    removing it makes it disappear."""
    from torchpq_amd.index import IVFPQIndex
    g = torch.Generator(device=DEV)
    g.manual_seed(2)
    d, n, nq = 128, 1_000_000, 4096
    centers = torch.rand(d, 256, generator=g, device=DEV) * 120
    base = (centers[:, torch.randint(0, 256, (n,), generator=g, device=DEV)]
            + torch.randn(d, n, generator=g, device=DEV) * 25).abs().round().contiguous()
    np.random.seed(2)
    idx = IVFPQIndex(d_vector=d, n_subvectors=64, n_cells=1024, initial_size=2048, device=DEV)
    idx.train(base[:, :100_000].contiguous())
    ids = idx.add(base)
    assert idx.n_items == n and torch.equal(ids, torch.arange(n, device=DEV))
    idx.n_probe = 32
    idx.use_smart_probing = False
    q = base[:, :nq].contiguous()
    v, i = idx.search(q, k=100)
    assert bool((v[:, 1:] <= v[:, :-1]).all())
    self_hit = (i == torch.arange(nq, device=DEV)[:, None]).any(1)
    assert float(self_hit.float().mean()) > 0.99
    # value of the self hit == -|x - decode(encode(x))|^2 (ADC identity), to fp32 accuracy
    codes = idx.encode(q)
    recon = idx.decode(codes)
    exact = -((q - recon) ** 2).sum(0)
    pos = (i == torch.arange(nq, device=DEV)[:, None]).float().argmax(1)
    got = v.gather(1, pos[:, None])[:, 0]
    rel = ((got - exact).abs() / exact.abs().clamp(min=1))[self_hit]
    assert float(rel.max()) < 1e-4
    idx.remove(ids=torch.arange(0, nq, 2, device=DEV))
    v2, i2 = idx.search(q, k=100)
    assert not bool((i2[0::2] == torch.arange(0, nq, 2, device=DEV)[:, None]).any())
    assert bool((i2[1::2] == torch.arange(1, nq, 2, device=DEV)[:, None]).any(1).float().mean() > 0.99)
