"""GPU: tpq_coarse_assign on wide vectors (128 < d <= 1024: csrc/lloyd.hip, "Wide vectors").

The coarse assign of IVFPQIndex.add / KMeans.predict (torchpq/clustering/KMeans.py:440-452 ->
kernels/MaxSimCuda.py:296-340, max_sim.cu:182-309) at descriptor widths beyond SIFT's (GIST: 960).
Bar: labels == tpq_max_sim's == the oracle's fp32 arg-max BIT FOR BIT, whatever the data -- the fp16
selection only decides a point when its bound allows it, every other point gets the exact kernel's own
value for each of its candidates (or, when the candidate lists overflow, the exact kernel itself).
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from test_gpu_lloyd import _data, T, N

pytestmark = pytest.mark.gpu

SHAPES = [(129, 1000, 300), (960, 3000, 700), (200, 5000, 2053), (1024, 777, 257), (300, 260, 1), (512, 100, 3000),
          (144, 20000, 512)]


@pytest.mark.parametrize("kind", ["gauss", "sift", "offset", "tiny", "huge", "heavy", "clusters"])
@pytest.mark.parametrize("shape", SHAPES)
def test_wide_assign_labels_are_the_fp32_arg_max(kind, shape, monkeypatch):
    import torchpq_amd.kernels as K
    monkeypatch.setattr(K.CoarseAssignHip, "default_route", "cascade")  # (small problems take other paths by default)
    d, m, n = shape
    rng = np.random.default_rng(hash((kind,) + shape) % 2 ** 31)
    x, cent = _data(kind, 1, d, m, min(n, m), rng)
    if n > m:
        extra = cent[:, :, rng.integers(0, cent.shape[2], n - m)] * (1 + 1e-3 * rng.standard_normal((1, d, n - m)))
        cent = np.concatenate([cent, extra.astype(np.float32)], axis=2)
    A, B = T(x[0]), T(cent[0])
    assert K.CoarseAssignHip.supported(d, m, n)
    op = K.CoarseAssignHip(distance="euclidean")
    vals, lab = op(A, B, return_vals=True)
    v32, l32 = K.MaxSimHip(distance="euclidean")(A, B, dim=1)
    assert torch.equal(lab, l32)
    if m * n * d < 3e9:
        _, el = c_oracle.max_sim(x, cent, "euclidean", "expanded")
        assert np.array_equal(N(lab), el[0])
    assert 0 <= op.last_rechecked() <= m
    # maxima: the exact kernel's for the re-checked points, the fast values (within the bound) otherwise
    scale = float((A.double().pow(2).sum(0).max() + B.double().pow(2).sum(0).max()).item())
    assert float((vals - v32).abs().max().item()) <= 2e-3 * scale


@pytest.mark.parametrize("shape", [(130, 1500, 300), (960, 2000, 513)])
def test_wide_assign_inner_product(shape, monkeypatch):
    import torchpq_amd.kernels as K
    monkeypatch.setattr(K.CoarseAssignHip, "default_route", "cascade")  # (small problems take other paths by default)
    d, m, n = shape
    rng = np.random.default_rng(11)
    x = (rng.standard_normal((d, m)) + 0.3).astype(np.float32)
    cent = x[:, rng.permutation(m)[:n]].copy()
    op = K.CoarseAssignHip(distance="inner")
    lab = op(T(x), T(cent))
    _, l32 = K.MaxSimHip(distance="inner")(T(x), T(cent), dim=1)
    assert torch.equal(lab, l32)
    _, el = c_oracle.max_sim(x[None], cent[None], "inner", "expanded")
    assert np.array_equal(N(lab), el[0])


def test_wide_assign_ties_overflow_and_flags(monkeypatch):
    """exact ties (duplicated centroids: the smaller index wins), candidate lists that overflow (every
    centroid a candidate of every point: the exact kernel takes over), a NaN point and an out-of-range
    centroid (flagged: everything exact)"""
    import torchpq_amd.kernels as K
    monkeypatch.setattr(K.CoarseAssignHip, "default_route", "cascade")  # (small problems take other paths by default)
    rng = np.random.default_rng(5)
    d, m, n = 160, 3000, 1024
    x = rng.integers(-9, 9, (d, m)).astype(np.float32)
    cent = x[:, :n].copy()
    cent[:, 512:] = cent[:, :512]                      # centroid blocks 2, 3 repeat blocks 0, 1
    op = K.CoarseAssignHip(distance="euclidean")
    lab = op(T(x), T(cent))
    _, el = c_oracle.max_sim(x[None], cent[None], "euclidean", "expanded")
    assert np.array_equal(N(lab), el[0]) and N(lab).max() < 512
    assert op.last_rechecked() == m                    # every point has a tie for its best
    cent_same = np.repeat(x[:, :1], 2000, axis=1)      # 2 000 identical centroids: 6 M candidate pairs
    lab = op(T(x), T(cent_same))
    assert int(lab.abs().max().item()) == 0
    zeros = np.zeros((d, 700), np.float32)             # all-zero data and centroids: bound = 0
    lab = op(T(zeros), T(zeros[:, :300]))
    assert int(lab.abs().max().item()) == 0
    x2 = (rng.standard_normal((d, m)) * 3).astype(np.float32)
    c2 = x2[:, :n].copy()
    x2[3, 17] = np.nan
    lab = op(T(x2), T(c2))
    _, l32 = K.MaxSimHip(distance="euclidean")(T(x2), T(c2), dim=1)
    ok = np.ones(m, bool)
    ok[17] = False                                      # (the NaN point's own label is arbitrary in both)
    assert np.array_equal(N(lab)[ok], N(l32)[ok])
    x3 = (rng.standard_normal((d, m)) * 3).astype(np.float32)
    c3 = x3[:, :n].copy()
    c3[:, 700] = 1.0e8
    lab = op(T(x3), T(c3))
    _, l32 = K.MaxSimHip(distance="euclidean")(T(x3), T(c3), dim=1)
    assert torch.equal(lab, l32)


def test_wide_small_problems_take_the_fp32_kernel_and_predict_uses_the_path():
    import torchpq_amd.kernels as K
    from torchpq_amd.clustering import KMeans
    rng = np.random.default_rng(2)
    d, m, n = 960, 300, 70
    x = rng.standard_normal((d, m)).astype(np.float32)
    cent = x[:, :n].copy()
    assert K.CoarseAssignHip.supported(d, m, n) and K.CoarseAssignHip.supported(960, 10 ** 6, 16384)
    assert not K.CoarseAssignHip.supported(1025, m, n)
    op = K.CoarseAssignHip(distance="euclidean")
    lab = op(T(x), T(cent))                            # default threshold: tpq_max_sim inside
    _, l32 = K.MaxSimHip(distance="euclidean")(T(x), T(cent), dim=1)
    assert torch.equal(lab, l32) and op.last_rechecked() == 0
    km = KMeans(n_clusters=256, distance="euclidean", max_iter=2, verbose=0)
    data = T(rng.standard_normal((384, 6000)).astype(np.float32))
    km.fit(data)
    q = T(rng.standard_normal((384, 4000)).astype(np.float32))
    km.fast_predict_min_work = 1
    fast = km.predict(q)
    _, ref = km.get_labels(q, km.centroids)
    assert torch.equal(fast, ref)


@pytest.mark.parametrize("shape", [(64, 5644, 1905), (200, 3000, 600), (128, 4000, 300), (960, 1500, 520)])
def test_centroids_beyond_the_fp16_range_of_the_data_scale_go_exact(shape, monkeypatch):
    """data with a spread of 1 around 1000, centroids with a spread of 50: the fp16 scale is set by the data, EVERY
    centroid overflows it (keys inf / NaN, threshold undefined) -- the problem is flagged and the whole list must
    go to the exact kernel, on the candidate routes of the narrow (chunked) and the wide path alike
    (found by tools/selection_soak.py --mode cascade: the candidate pass emitted nothing and no fallback ran)"""
    import torchpq_amd.kernels as K
    monkeypatch.setattr(K.CoarseAssignHip, "default_route", "cascade")  # (small problems take other paths by default)
    d, m, n = shape
    rng = np.random.default_rng(d + m)
    x = (1000.0 + rng.standard_normal((d, m))).astype(np.float32)
    cent = (x[:, rng.integers(0, m, n)] + 50.0 * rng.standard_normal((d, n))).astype(np.float32)
    op = K.CoarseAssignHip(distance="euclidean")
    lab = op(T(x), T(cent))
    _, l32 = K.MaxSimHip(distance="euclidean")(T(x), T(cent), dim=1)
    assert torch.equal(lab, l32) and int(lab.max()) < n
    assert op.last_rechecked() == m
