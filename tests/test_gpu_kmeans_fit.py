"""GPU parity of the Lloyd DRIVER: torchpq_amd MultiKMeans.fit / KMeans.fit
against oracle.kmeans_fit_redo and against the reference's own MultiKMeans.fit run on CPU
(tests/golden/fx_kmeans_fit.npz, made by make_golden.py::fx_kmeans_fit).

Reference: clustering/MultiKMeans.py:415-453 (steps, `error <= tol` exit, best-inertia redo,
labels of the last assign), KMeans.py:399-438, initialize_centroids :270-289.
Tolerances: labels equal except where the two best centroids are within 1e-4 relative of each
other (near-ties: the update sums in a different fp32 order than the oracle's fp64), centroids
1e-5 relative (+ 1e-4 absolute on values up to 218).
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


CASES = {  # name -> (init key, n_redo, max_iter, tol key or value, seed key)
    "1": ("init", 1, 1, 0.0, None),
    "3": ("init", 1, 3, 0.0, None),
    "tol": ("init", 1, 12, "tol_exit", None),
    "redo": ("init", 2, 3, 0.0, "redo_seed"),
    "redo_b": ("bad_init", 2, 3, 0.0, "redo_b_seed"),
}


def _run_gpu(fx, case, single=False, precision="bf16x3"):
    from torchpq_amd.clustering import KMeans, MultiKMeans
    init_key, n_redo, max_iter, tol, seed = CASES[case]
    tol = float(fx[tol]) if isinstance(tol, str) else tol
    init = fx[init_key]
    if seed is not None:
        np.random.seed(int(fx[seed]))
    if single:
        km = KMeans(n_clusters=init.shape[2], n_redo=n_redo, max_iter=max_iter, tol=tol,
                    assign_precision=precision)
        labels = km.fit(T(fx["data"][0]), T(init[0]))
        return N(km.centroids)[None], N(labels)[None]
    mk = MultiKMeans(n_clusters=init.shape[2], n_redo=n_redo, max_iter=max_iter, tol=tol,
                     assign_precision=precision)
    labels = mk.fit(T(fx["data"]), T(init))
    return N(mk.centroids), N(labels)


def _near_tie_ok(data, centroids_before_last_assign, got, exp):
    """label disagreements only where the two best similarities are within 1e-4 relative"""
    bad = np.argwhere(got != exp)
    for b, i in bad:
        x = data[b, :, i].astype(np.float64)
        s = -((centroids_before_last_assign[b].astype(np.float64) - x[:, None]) ** 2).sum(0)
        if abs(s[got[b, i]] - s[exp[b, i]]) > 1e-4 * abs(s.max()) + 1e-6:
            return False
    return True


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", list(CASES))
def test_multikmeans_fit_vs_oracle_and_reference(fx_kmeans_fit, case, precision, monkeypatch):
    """both arithmetic modes of the Lloyd loop's assign step ("bf16x3" is the default; the fixture's
    d = 8 lies below MultiKMeans.split_min_d, so the threshold is lowered to run the split kernel)"""
    from torchpq_amd.clustering import MultiKMeans
    monkeypatch.setattr(MultiKMeans, "split_min_d", 1)
    fx = fx_kmeans_fit
    cen, labels = _run_gpu(fx, case, precision=precision)
    init_key, n_redo, max_iter, tol, seed = CASES[case]
    tol = float(fx[tol]) if isinstance(tol, str) else tol
    if seed is not None:
        np.random.seed(int(fx[seed]))
    o_cen, o_lab, o_inert, o_steps = orc.kmeans_fit_redo(
        fx["data"], fx[init_key].copy(), n_redo, max_iter, tol, fx[init_key].shape[2],
        assign=lambda a, b: c_oracle.max_sim(a, b, "euclidean", "expanded"))
    for exp_cen, exp_lab, who in ((o_cen, o_lab, "oracle"),
                                  (fx[f"ref_centroids_{case}"], fx[f"ref_labels_{case}"], "reference")):
        assert labels.shape == exp_lab.shape and labels.dtype == np.int64
        assert (labels == exp_lab).mean() >= 0.999, who
        np.testing.assert_allclose(cen, exp_cen, rtol=1e-5, atol=1e-4, err_msg=who)


def test_fit_stops_at_the_tolerance_step(fx_kmeans_fit):
    """`error <= tol` leaves the loop after exactly the step the reference stops at: the result
    equals the 4-step run and differs from the 3- and 5-step runs"""
    from torchpq_amd.clustering import MultiKMeans
    fx = fx_kmeans_fit
    k = fx["init"].shape[2]
    runs = {}
    for steps in (3, 4, 5):
        mk = MultiKMeans(n_clusters=k, max_iter=steps, tol=0.0)
        mk.fit(T(fx["data"]), T(fx["init"]))
        runs[steps] = N(mk.centroids)
    mk = MultiKMeans(n_clusters=k, max_iter=12, tol=float(fx["tol_exit"]))
    mk.fit(T(fx["data"]), T(fx["init"]))
    got = N(mk.centroids)
    assert int(fx["tol_exit_steps"]) == 4
    assert np.array_equal(got, runs[4])
    assert not np.array_equal(got, runs[3]) and not np.array_equal(got, runs[5])


def test_redo_keeps_the_best_inertia(fx_kmeans_fit):
    """n_redo=2: from the good start redo 0 wins, from the poor start redo 1 (np.random init)
    wins -- as in the reference run; the registered centroids are the winner's"""
    from torchpq_amd.clustering import MultiKMeans
    fx = fx_kmeans_fit
    k = fx["init"].shape[2]
    for case, winner in (("redo", 0), ("redo_b", 1)):
        init = fx[CASES[case][0]]
        assert int(np.argmin(fx[f"{'redo' if case == 'redo' else 'redo_b'}_inertia"])) == winner
        cen, labels = _run_gpu(fx, case)
        # single-redo runs of both starts, to identify the winner independently
        mk0 = MultiKMeans(n_clusters=k, max_iter=3, tol=0.0)
        mk0.fit(T(fx["data"]), T(init))
        np.random.seed(int(fx[CASES[case][4]]))
        mk1 = MultiKMeans(n_clusters=k, max_iter=3, tol=0.0)
        mk1.fit(T(fx["data"]))
        assert np.array_equal(cen, N((mk0, mk1)[winner].centroids))
        assert not np.array_equal(cen, N((mk0, mk1)[1 - winner].centroids))


@pytest.mark.parametrize("case", ["3", "tol", "redo_b"])
def test_kmeans_single_problem_fit(fx_kmeans_fit, case):
    """KMeans.fit (KMeans.py:399-438) == sub-problem 0 of the batched driver, vs the oracle"""
    fx = fx_kmeans_fit
    init_key, n_redo, max_iter, tol, seed = CASES[case]
    tol = float(fx[tol]) if isinstance(tol, str) else tol
    cen, labels = _run_gpu(fx, case, single=True)
    if seed is not None:
        np.random.seed(int(fx[seed]))
    data0 = fx["data"][:1]
    o_cen, o_lab, _, _ = orc.kmeans_fit_redo(
        data0, fx[init_key][:1].copy(), n_redo, max_iter, tol, fx[init_key].shape[2],
        assign=lambda a, b: c_oracle.max_sim(a, b, "euclidean", "expanded"))
    assert (labels == o_lab).mean() >= 0.999
    np.testing.assert_allclose(cen, o_cen, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("l,d,n,k,steps", [(2, 16, 5000, 256, 3), (1, 64, 4000, 64, 2),
                                           (4, 2, 20000, 256, 3), (2, 40, 3000, 300, 2)])
def test_fit_random_data_vs_oracle(l, d, n, k, steps, precision):
    """Gaussian (signed, non-integer) data, codebook-sized problems (the MFMA update path):
    labels of the last assign equal the oracle's except at near-ties, centroids 1e-5"""
    from torchpq_amd.clustering import MultiKMeans
    rng = np.random.default_rng(l * 100 + d)
    data = rng.standard_normal((l, d, n)).astype(np.float32) * 3
    init = data[:, :, rng.choice(n, k, replace=False)].copy()
    mk = MultiKMeans(n_clusters=k, max_iter=steps, tol=0.0, assign_precision=precision)
    used = mk._assign_path(l, d, n, k, training=True)
    want = "fp32" if precision == "fp32" or d < mk.split_min_d else ("select" if k <= 256 else "bf16x3")
    assert used == want and mk._assign_path(l, d, n, k, training=False) == "fp32"
    labels = N(mk.fit(T(data), T(init)))
    cen = N(mk.centroids)
    o_cen, o_lab, _ = _oracle_fit(data, init, steps)
    prev = _oracle_fit(data, init, steps - 1)[0] if steps > 1 else init
    assert (labels == o_lab).mean() >= 0.998
    assert _near_tie_ok(data, prev, labels, o_lab)
    # centroids of clusters whose membership is identical agree to fp32 summation error
    same_members = np.ones((l, k), bool)
    for b, i in np.argwhere(labels != o_lab):
        same_members[b, labels[b, i]] = same_members[b, o_lab[b, i]] = False
    np.testing.assert_allclose(cen.transpose(0, 2, 1)[same_members],
                               o_cen.transpose(0, 2, 1)[same_members], rtol=1e-5, atol=1e-5)


def _oracle_fit(data, init, steps):
    cen, lab, _, _ = orc.kmeans_fit_redo(
        data, init.copy(), 1, steps, 0.0, init.shape[2],
        assign=lambda a, b: c_oracle.max_sim(a, b, "euclidean", "expanded"))
    return cen, lab, None


def test_kmeans_public_helpers():
    """the reference's public helper surface on KMeans / MultiKMeans (KMeans.py:117-283): sim
    helpers never mutate their inputs (the reference's in-place CPU forms do), kmeans++ seeding
    returns data points, memory helpers answer"""
    from torchpq_amd.clustering import KMeans, MultiKMeans
    rng = np.random.default_rng(4)
    a = T(rng.standard_normal((16, 300)).astype(np.float32))
    b = T(rng.standard_normal((16, 20)).astype(np.float32))
    a0, b0 = a.clone(), b.clone()
    e = KMeans.euc_sim(a, b)
    c = KMeans.cos_sim(a, b)
    assert torch.equal(a, a0) and torch.equal(b, b0)
    ref = -((N(a).T[:, None, :] - N(b).T[None, :, :]) ** 2).sum(-1)
    np.testing.assert_allclose(N(e), ref, rtol=1e-4, atol=1e-4)
    an, bn = N(a) / np.linalg.norm(N(a), axis=0), N(b) / np.linalg.norm(N(b), axis=0)
    np.testing.assert_allclose(N(c), an.T @ bn, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(N(MultiKMeans.euc_sim(a[None], b[None]))[0], ref, rtol=1e-4, atol=1e-4)
    np.random.seed(1)
    km = KMeans(n_clusters=8, init_mode="kmeans++", max_iter=2)
    seeds = km.kmeanspp(a)
    assert seeds.shape == (16, 8)
    cols = N(a).T
    assert all(any(np.array_equal(s, col) for col in cols) for s in N(seeds).T)   # seeds are data points
    labels = km.fit(a)
    assert labels.shape == (300,) and km.centroids.shape == (16, 8)
    assert KMeans.remaining_memory("cuda:0") > 0 and MultiKMeans.does_it_fit((4, 4), device="cuda:0")
    assert not MultiKMeans.does_it_fit((1 << 40,), device="cuda:0")
    km.warmup_kernels()


def test_single_problem_training_assign_takes_exact_labels_from_the_coarse_assign(monkeypatch):
    """l = 1, many centroids (the coarse quantiser): the Lloyd loop's labels come from
    tpq_coarse_assign and are the fp32 kernel's, bit for bit, at every step of a fit (compared on
    shared centroids: the update's float atomics make two runs of the SAME fit differ in the last
    bits of the centroids); the maxima it reports are the selection's fast values, within 1e-4."""
    from torchpq_amd import kernels as K
    from torchpq_amd.clustering import MultiKMeans
    monkeypatch.setattr(MultiKMeans, "coarse_min_work", 0)
    rng = np.random.default_rng(21)
    d, n, k = 96, 20000, 300
    centers = rng.standard_normal((d, 40)) * 6
    data = (centers[:, rng.integers(0, 40, n)] + rng.standard_normal((d, n))).astype(np.float32)
    cen = T(data[:, rng.choice(n, k, replace=False)].copy()[None])
    calls = []
    orig = K.CoarseAssignHip.__call__
    monkeypatch.setattr(K.CoarseAssignHip, "__call__",
                        lambda self, A, B, **kw: (calls.append(kw), orig(self, A, B, **kw))[1])
    fast = MultiKMeans(n_clusters=k, assign_precision="bf16x3")
    exact = MultiKMeans(n_clusters=k, assign_precision="fp32")
    x = T(data[None])
    for step in range(4):
        v1, l1 = fast.get_labels(x, cen, training=True)
        v0, l0 = exact.get_labels(x, cen, training=True)
        assert torch.equal(l1, l0), step
        np.testing.assert_allclose(N(v1), N(v0), rtol=1e-4, atol=1e-3 * float(np.abs(N(v0)).max()))
        cen = exact.compute_centroids(x, l0)
    assert len(calls) == 4 and all(c.get("return_vals") for c in calls)
    # outside the Lloyd loop (predict, k-means++) the batched engine keeps the fp32 kernel
    fast.get_labels(x, cen)
    assert len(calls) == 4
