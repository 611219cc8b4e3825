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
