"""GPU: one Lloyd iteration on prepared data (tpq_lloyd_prepare / tpq_lloyd_step, csrc/lloyd.hip).

The step replaces the get_labels -> compute_centroids pair of the reference's driver
(torchpq/clustering/MultiKMeans.py:415-453; max_sim_tn kernels/cuda/max_sim.cu:182-309;
compute_centroids kernels/cuda/compute_centroids.cu:10-86).  Bar: labels == the fp32 kernel's and the
oracle's fp32 arg-max BIT FOR BIT whatever the data (the fp16 cascade only ever decides a point when
its rigorous bound allows it); new centroids == tpq_compute_centroids of those labels; the fit()
driver through this path reproduces the oracle's and the reference's own runs.
"""
import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import ivfpq_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


def _data(kind, l, d, n, k, rng):
    if kind == "gauss":
        x = rng.standard_normal((l, d, n)) * 3
    elif kind == "sift":          # small non-negative integers: exact in every arithmetic
        x = rng.integers(0, 219, (l, d, n)).astype(np.float64)
    elif kind == "offset":        # a large common offset: centring matters
        x = rng.standard_normal((l, d, n)) + 1.0e4
    elif kind == "tiny":
        x = rng.standard_normal((l, d, n)) * 1e-18
    elif kind == "huge":
        x = rng.standard_normal((l, d, n)) * 1e14
    elif kind == "heavy":         # heavy tails: a few elements dominate max |x - mu| (the fp16 scale)
        x = rng.standard_t(1.5, (l, d, n))
    elif kind == "clusters":      # tight clusters: near-ties between neighbouring centroids are common
        c = rng.standard_normal((l, d, 40)) * 5
        x = c[:, :, rng.integers(0, 40, n)] + rng.standard_normal((l, d, n)) * 0.05
    else:
        raise ValueError(kind)
    x = x.astype(np.float32)
    cent = x[:, :, rng.permutation(n)[:k]].copy()
    return x, cent


SHAPES = [(3, 64, 4097, 256), (2, 40, 3001, 200), (5, 16, 1000, 256), (1, 33, 777, 17), (4, 12, 65, 3),
          (2, 64, 64, 256), (1, 1, 500, 7)]


@pytest.mark.parametrize("kind", ["gauss", "sift", "offset", "tiny", "huge", "heavy", "clusters"])
@pytest.mark.parametrize("shape", SHAPES[:5])
def test_lloyd_step_labels_are_the_fp32_arg_max(kind, shape):
    import torchpq_amd.kernels as K
    l, d, n, k = shape
    rng = np.random.default_rng(hash((kind, shape)) % 2 ** 31)
    x, cent = _data(kind, l, d, n, k, rng)
    step = K.LloydStepHip(T(x), T(cent))
    vals, lab, new = step(T(cent))
    v32, l32 = K.MaxSimHip(distance="euclidean")(T(x), T(cent), dim=2, mode="tn")
    assert torch.equal(lab, l32)
    ev, el = c_oracle.max_sim(x, cent, "euclidean", "expanded")
    assert np.array_equal(N(lab), el)
    ref_new = K.ComputeCentroidsHip()(T(x), lab, k=k)
    scale = float(ref_new.abs().max()) + 1e-30
    assert float((new - ref_new).abs().max()) <= 2e-6 * scale
    # the maxima: fast values for decided points -- the coarse level's, within its bound: ~5e-4 of
    # (|x - mu| + |c - mu|)^2, mu = the mean initial centroid -- exact for re-checked points
    mu = cent.astype(np.float64).mean(axis=2, keepdims=True)
    an2 = ((x - mu) ** 2).sum(axis=1).max() + ((cent - mu) ** 2).sum(axis=1).max()
    assert np.abs(N(vals).astype(np.float64) - ev).max() <= 2e-3 * 2 * an2 + 1e-30
    # a second iteration from the new centroids (the prepared block keeps serving)
    vals2, lab2, _ = step(new)
    _, l32b = K.MaxSimHip(distance="euclidean")(T(x), new, dim=2, mode="tn")
    assert torch.equal(lab2, l32b)


@pytest.mark.parametrize("shape", SHAPES[5:])
def test_lloyd_step_degenerate_shapes(shape):
    import torchpq_amd.kernels as K
    l, d, n, k = shape
    rng = np.random.default_rng(7)
    x = (rng.standard_normal((l, d, n)) * 2).astype(np.float32)
    cent = (rng.standard_normal((l, d, k)) * 2).astype(np.float32)   # more centroids than points is fine
    step = K.LloydStepHip(T(x), T(cent))
    _, lab, new = step(T(cent))
    _, l32 = K.MaxSimHip(distance="euclidean")(T(x), T(cent), dim=2, mode="tn")
    assert torch.equal(lab, l32)
    ref_new = K.ComputeCentroidsHip()(T(x), lab, k=k)
    assert float((new - ref_new).abs().max()) <= 2e-6 * float(ref_new.abs().max() + 1e-30)


def test_lloyd_step_exact_ties_duplicates_and_points_on_centroids():
    """duplicated centroids (every value tie goes to the smaller index), points equal to centroids,
    a codebook whose entries differ by single ulps: everything ambiguous ends in the exact kernel"""
    import torchpq_amd.kernels as K
    rng = np.random.default_rng(11)
    l, d, n, k = 2, 32, 5000, 128
    x = rng.integers(-20, 20, (l, d, n)).astype(np.float32)
    cent = x[:, :, :k].copy()
    cent[:, :, k // 2:] = cent[:, :, :k // 2]            # every centroid twice
    step = K.LloydStepHip(T(x), T(cent))
    _, lab, _ = step(T(cent))
    _, el = c_oracle.max_sim(x, cent, "euclidean", "expanded")
    assert np.array_equal(N(lab), el) and N(lab).max() < k // 2
    assert float(step.rechecked().sum()) == l * n          # b1 == b2 everywhere
    crowd = np.repeat(x[:, :, :1], k, axis=2).copy()
    crowd[:, 0, :] = np.nextafter(crowd[:, 0, 0:1], np.inf) + np.arange(k, dtype=np.float32) * 0
    crowd[:, 0, :] = crowd[:, 0, :] * (1 + np.arange(k, dtype=np.float32) * 2.0 ** -22)
    _, lab, _ = step(T(crowd))
    _, el = c_oracle.max_sim(x, crowd, "euclidean", "expanded")
    assert np.array_equal(N(lab), el)


def test_lloyd_step_non_finite_data_and_centroids_go_through_the_exact_path():
    """a NaN / Inf anywhere in a sub-problem's data, or centroids beyond the fp16 range of the scaled
    problem, flag the sub-problem: its points are all re-checked by the fp32 kernel; the other
    sub-problems keep the fast path; labels equal the fp32 kernel's in both"""
    import torchpq_amd.kernels as K
    rng = np.random.default_rng(13)
    l, d, n, k = 3, 24, 2000, 64
    x = (rng.standard_normal((l, d, n)) * 2).astype(np.float32)
    cent = x[:, :, :k].copy()
    x[1, 3, 77] = np.nan
    x[1, 5, 78] = np.inf
    step = K.LloydStepHip(T(x), T(cent))
    _, lab, _ = step(T(cent))
    _, l32 = K.MaxSimHip(distance="euclidean")(T(x), T(cent), dim=2, mode="tn")
    assert torch.equal(lab, l32)
    rc = N(step.rechecked())
    assert rc[1] == n and rc[0] < n // 10 and rc[2] < n // 10
    far = cent.copy()
    far[2, :, 5] = 1.0e9                                   # 2 s |c - mu| overflows fp16 in sub-problem 2
    _, lab, _ = step(T(far))
    _, l32 = K.MaxSimHip(distance="euclidean")(T(x), T(far), dim=2, mode="tn")
    assert torch.equal(lab, l32)
    assert N(step.rechecked())[2] == n


def test_lloyd_prepare_scale_from_a_sample_flags_what_the_sample_missed():
    """from 2^18 points on tpq_lloyd_prepare reads every sixteenth 4-KiB run of a row for the scale and leaves one
    bit of headroom: an outlier in an unsampled run that exceeds it flags its sub-problem (all of it re-checked
    exactly); one inside the headroom does not; labels equal the fp32 kernel's either way"""
    import torchpq_amd.kernels as K
    rng = np.random.default_rng(17)
    l, d, n, k = 3, 16, 300_000, 64
    x = rng.standard_normal((l, d, n)).astype(np.float32)
    cent = x[:, :, :k].copy()
    big = float(np.abs(x).max())
    x[1, 4, 5000] = 9.0 * big        # run 4 of its row (5000 // 1024): not a sampled run, beyond the headroom
    x[2, 7, 7000] = 1.5 * big        # unsampled too, inside the headroom
    step = K.LloydStepHip(T(x), T(cent))
    _, lab, new = step(T(cent))
    _, l32 = K.MaxSimHip(distance="euclidean")(T(x), T(cent), dim=2, mode="tn")
    assert torch.equal(lab, l32)
    rc = N(step.rechecked())
    assert rc[1] == n and rc[0] < n // 5 and rc[2] < n // 5
    ref_new = K.ComputeCentroidsHip()(T(x), l32, k)
    assert float((new - ref_new).abs().max()) <= 1e-5 * float(ref_new.abs().max())


@pytest.mark.parametrize("case", ["1", "3", "tol", "redo", "redo_b"])
def test_fit_through_the_prepared_path_vs_oracle_and_reference(fx_kmeans_fit, case, monkeypatch):
    """MultiKMeans.fit forced through LloydStepHip on the reference-made fixture (d = 8: the gate on
    split_min_d is lowered): same bar as the per-kernel path in tests/test_gpu_kmeans_fit.py"""
    from torchpq_amd.clustering import MultiKMeans
    from tests_support import CASES
    monkeypatch.setattr(MultiKMeans, "split_min_d", 1)
    monkeypatch.setattr(MultiKMeans, "lloyd_min_work", 0)
    monkeypatch.setattr(MultiKMeans, "lloyd_min_iter", 1)
    import torchpq_amd.kernels as K
    calls = []
    orig = K.LloydStepHip.__call__
    monkeypatch.setattr(K.LloydStepHip, "__call__", lambda self, *a, **kw: (calls.append(1), orig(self, *a, **kw))[1])
    fx = fx_kmeans_fit
    init_key, n_redo, max_iter, tol, seed = CASES[case]
    tol = float(fx[tol]) if isinstance(tol, str) else tol
    if seed is not None:
        np.random.seed(int(fx[seed]))
    mk = MultiKMeans(n_clusters=fx[init_key].shape[2], n_redo=n_redo, max_iter=max_iter, tol=tol)
    labels = N(mk.fit(T(fx["data"]), T(fx[init_key])))
    cen = N(mk.centroids)
    assert len(calls) >= 1
    if seed is not None:
        np.random.seed(int(fx[seed]))
    o_cen, o_lab, _, _ = orc.kmeans_fit_redo(
        fx["data"], fx[init_key].copy(), n_redo, max_iter, tol, fx[init_key].shape[2],
        assign=lambda a, b: c_oracle.max_sim(a, b, "euclidean", "expanded"))
    for exp_cen, exp_lab, who in ((o_cen, o_lab, "oracle"),
                                  (fx[f"ref_centroids_{case}"], fx[f"ref_labels_{case}"], "reference")):
        assert (labels == exp_lab).mean() >= 0.999, who
        np.testing.assert_allclose(cen, exp_cen, rtol=1e-5, atol=1e-4, err_msg=who)


def test_fit_prepared_path_equals_per_kernel_path_on_wide_problems():
    """d = 64, k = 256 (the PQ-codebook training shape, reduced n): the two paths of fit() give the same
    labels at every step count and centroids equal to fp32 summation-order noise"""
    from torchpq_amd.clustering import MultiKMeans
    rng = np.random.default_rng(21)
    l, d, n, k = 4, 64, 30000, 256
    x = T((rng.standard_normal((l, d, n)) * 2).astype(np.float32))
    init = x[:, :, :k].contiguous()
    out = {}
    for name, work in (("prepared", 0), ("kernels", 1 << 62)):
        mk = MultiKMeans(n_clusters=k, max_iter=5, tol=0.0)
        mk.lloyd_min_work, mk.lloyd_min_iter = work, 1
        out[name] = (mk.fit(x, init.clone()), mk.centroids)
    agree = float((out["prepared"][0] == out["kernels"][0]).double().mean())
    assert agree > 0.9995, agree            # (5 iterations amplify last-bit differences of the update)
    assert float((out["prepared"][1] - out["kernels"][1]).abs().max()) < 1e-3


# ---------------------------------------------------------------------------------------------
# tpq_coarse_assign through the cascade (one problem, many centroids, d <= 128)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind", ["gauss", "sift", "offset", "huge", "heavy", "clusters"])
@pytest.mark.parametrize("d,m,n", [(128, 6000, 1000), (100, 3001, 777), (64, 5000, 4200), (33, 2000, 300),
                                   (128, 300, 5000), (7, 4097, 257), (128, 70000, 4096)])
def test_coarse_assign_cascade_labels_are_the_fp32_arg_max(kind, d, m, n, monkeypatch):
    """the fp16 cascade behind tpq_coarse_assign (forced on for every shape: by default it takes over
    from 4 096 centroids on): labels == tpq_max_sim == the C oracle, chunked (n > 256) or not"""
    import torchpq_amd.kernels as K
    monkeypatch.setattr(K.CoarseAssignHip, "default_route", "cascade")  # (small problems take other paths by default)
    rng = np.random.default_rng(hash((kind, d, m, n)) % 2 ** 31)
    x, cent = _data(kind, 1, d, m, min(n, m), rng)
    if n > m:  # more centroids than points: pad with perturbed copies
        extra = cent[:, :, rng.integers(0, cent.shape[2], n - m)] * (1 + 1e-3 * rng.standard_normal((1, d, n - m)))
        cent = np.concatenate([cent, extra.astype(np.float32)], axis=2)
    A, B = T(x[0]), T(cent[0])
    op = K.CoarseAssignHip(distance="euclidean")
    vals, lab = op(A, B, return_vals=True)
    _, l32 = K.MaxSimHip(distance="euclidean")(A, B, dim=1)
    assert torch.equal(lab, l32)
    if m * n * d < 3e9:
        _, el = c_oracle.max_sim(x, cent, "euclidean", "expanded")
        assert np.array_equal(N(lab), el[0])
    assert 0 <= op.last_rechecked() <= m


def test_coarse_assign_cascade_ties_and_flags(monkeypatch):
    """duplicated centroids across chunk boundaries (ties must go to the smaller index), a NaN point and
    an out-of-range centroid (the problem is flagged: everything re-checked exactly)"""
    import torchpq_amd.kernels as K
    monkeypatch.setattr(K.CoarseAssignHip, "default_route", "cascade")  # (small problems take other paths by default)
    rng = np.random.default_rng(5)
    d, m, n = 48, 3000, 1024
    x = rng.integers(-9, 9, (d, m)).astype(np.float32)
    cent = x[:, :n].copy()
    cent[:, 512:] = cent[:, :512]                      # chunk 2, 3 repeat chunk 0, 1
    op = K.CoarseAssignHip(distance="euclidean")
    lab = op(T(x), T(cent))
    _, el = c_oracle.max_sim(x[None], cent[None], "euclidean", "expanded")
    assert np.array_equal(N(lab), el[0]) and N(lab).max() < 512
    assert op.last_rechecked() == m
    x2 = (rng.standard_normal((d, m)) * 3).astype(np.float32)
    c2 = x2[:, :n].copy()
    x2[3, 17] = np.nan
    lab = op(T(x2), T(c2))
    _, l32 = K.MaxSimHip(distance="euclidean")(T(x2), T(c2), dim=1)
    assert torch.equal(lab, l32) and op.last_rechecked() == m
    x3 = (rng.standard_normal((d, m)) * 3).astype(np.float32)
    c3 = x3[:, :n].copy()
    c3[:, 700] = 1.0e8
    lab = op(T(x3), T(c3))
    _, l32 = K.MaxSimHip(distance="euclidean")(T(x3), T(c3), dim=1)
    assert torch.equal(lab, l32) and op.last_rechecked() == m
