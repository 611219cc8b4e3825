"""Pin the oracle against the reference ITSELF, executed here (build container only).

TEST INFRASTRUCTURE.  Imports DeMoriarty/TorchPQ from /root/reference through oracle/_refimport.py
(stub cupy: only the reference's own CPU / PyTorch code paths can run) and checks every oracle
function that has a runnable counterpart, on fresh random inputs -- independently of the committed
golden vectors (tests/golden/, which were produced by the same reference and travel to the GPU box).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.pin_against_reference        # prints one line per check

tests/test_oracle_golden.py::test_oracle_pinned_against_live_reference runs it when the reference
tree is present and skips otherwise.  What cannot be pinned this way -- the list scan, which has no
CPU implementation in the reference -- is pinned indirectly (identity with the reference's
precompute_adc / decode / metric, see ivfpq_oracle.py's header).
"""
from __future__ import annotations

import numpy as np

from . import ivfpq_oracle as orc
from ._refimport import available, import_reference


def _nonneg(rng, d, n):
    """non-negative data: the reference's CPU k-means paths abs() their inputs (KMeans.py:347)"""
    return np.abs(rng.standard_normal((d, n)) * 20).astype(np.float32)


def run(verbose=True):
    assert available(), "the reference tree is not present"
    tq = import_reference()
    import torch
    rng = np.random.default_rng(2024)
    done = []

    def ok(name):
        done.append(name)
        if verbose:
            print("pinned:", name)

    # coarse similarities (metric.py:31-98)
    a, b = _nonneg(rng, 48, 37), _nonneg(rng, 48, 29)
    ref = tq.metric.negative_squared_l2_distance(torch.from_numpy(a.copy()), torch.from_numpy(b.copy())).numpy()
    got = orc.neg_sq_l2(a, b)
    np.testing.assert_allclose(got, ref, rtol=1e-5, atol=1e-5 * np.abs(ref).max())
    ok("neg_sq_l2 == metric.negative_squared_l2_distance")

    # ADC look-up table (codec/PQCodec.py:62-75) on a codec with injected codebooks
    m, ds, nq = 8, 4, 19
    codec = tq.codec.PQCodec(d_vector=m * ds, n_subvectors=m, n_clusters=256, distance="euclidean")
    cb = (rng.standard_normal((m, ds, 256)) * 10).astype(np.float32)
    codec.kmeans.register_buffer("centroids", torch.from_numpy(cb.copy()))
    codec._is_trained = torch.tensor(True) if hasattr(codec, "_is_trained") else True
    q = (rng.standard_normal((m * ds, nq)) * 10).astype(np.float32)
    ref = codec.precompute_adc(torch.from_numpy(q.copy())).numpy()
    got = orc.adc_lut(q, cb, "euclidean")
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())
    ok("adc_lut == PQCodec.precompute_adc")

    # decode (codec/PQCodec.py:95-130)
    codes = rng.integers(0, 256, (m, 50), dtype=np.uint8)
    ref = codec.decode(torch.from_numpy(codes.copy())).numpy()
    assert np.array_equal(orc.pq_decode(cb, codes), ref)
    ok("pq_decode == PQCodec.decode")

    # smart probing (index/IVFPQIndex.py:499-512), the torch expression evaluated as the reference does
    n_probe = 16
    sims = -np.sort(np.abs(rng.standard_normal((200, n_probe)).astype(np.float32)) * 3e4, axis=1)
    t = torch.from_numpy(sims.copy())
    p = torch.softmax(-t.abs().sqrt() / 30.0, dim=-1)
    ne = -torch.sum(p * torch.log2(p) / torch.log2(torch.tensor(n_probe)), dim=-1)
    ref = torch.ceil(ne * n_probe).long().numpy()
    got = orc.smart_probing(sims, n_probe, 30.0)
    assert np.abs(got - ref).max() <= 1 and (got != ref).mean() < 0.01
    ok("smart_probing == IVFPQIndex.search's entropy rule")

    # container: add sequences with growth, look-ups (container/CellContainer.py:97-367)
    for mode, step in (("double", 8), ("step", 8)):
        c = tq.container.CellContainer(code_size=8, n_cells=6, dtype="uint8", device="cpu", initial_size=4,
                                       expand_step_size=step, expand_mode=mode,
                                       use_inverse_id_mapping=True, contiguous_size=4)
        o = orc.ContainerState(8, 6, 4, expand_step_size=step, expand_mode=mode)
        for nb in (5, 23, 2, 61):
            data = rng.integers(0, 256, (8, nb), dtype=np.uint8)
            cells = rng.integers(0, 6, nb).astype(np.int64)
            r_ids, r_adr = c.add(torch.from_numpy(data.copy()), torch.from_numpy(cells.copy()), return_address=True)
            assert np.array_equal(c.get_ioa(torch.from_numpy(cells.copy())).numpy(), orc.get_ioa(cells))
            o_ids, o_adr = o.add(data, cells)
            assert np.array_equal(r_ids.numpy(), o_ids) and np.array_equal(r_adr.numpy(), o_adr)
            for name, arr in (("_storage", o.storage), ("_cell_start", o.cell_start), ("_cell_size", o.cell_size),
                              ("_cell_capacity", o.cell_capacity), ("_is_empty", o.is_empty),
                              ("_address2id", o.address2id)):
                assert np.array_equal(getattr(c, name).numpy(), arr), (mode, nb, name)
        probe = np.arange(-2, c.capacity + 2).astype(np.int64)
        assert np.array_equal(c.get_cell_by_address(torch.from_numpy(probe.copy())).numpy(),
                              orc.get_cell_by_address(probe, o.cell_start, o.cell_capacity))
        assert np.array_equal(c.get_id_by_address(torch.from_numpy(probe.copy())).numpy(),
                              orc.get_id_by_address(o.address2id, probe))
        assert np.array_equal(c.get_data_by_address(torch.from_numpy(probe.copy())).numpy(),
                              orc.storage_to_codes(o.storage, probe))
    ok("ContainerState.add / expand / get_ioa / look-ups == CellContainer (double and step)")

    # k-means assign / update (clustering/MultiKMeans.py:334-392, CPU paths; last point unlabelled)
    l, d, n, k = 3, 6, 700, 16
    data = np.stack([_nonneg(rng, d, n) for _ in range(l)])
    init = data[:, :, rng.choice(n, k, replace=False)].copy()
    mk = tq.clustering.MultiKMeans(n_clusters=k, distance="euclidean", max_iter=1)
    _, lab = mk.get_labels(torch.from_numpy(data.copy()), torch.from_numpy(init.copy()))
    ref_lab = lab.numpy()[:, :-1]
    sims = mk.euc_sim(torch.from_numpy(data.copy()), torch.from_numpy(init.copy())).numpy()
    gap = np.sort(sims, -1)[..., -1] - np.sort(sims, -1)[..., -2]
    got_lab = orc.max_sim(data, init, "euclidean", "expanded")[1][:, :-1]
    same = got_lab == ref_lab
    # disagreements only where the two best centroids are (nearly) tied
    assert same.mean() > 0.99, same.mean()
    assert np.all(gap[:, :-1][~same] <= 1e-3 * np.abs(sims).max()), gap[:, :-1][~same]
    full = orc.max_sim(data, init, "euclidean", "expanded")[1]
    ref_c = mk.compute_centroids(torch.from_numpy(data.copy()), torch.from_numpy(full.copy())).numpy()
    np.testing.assert_allclose(orc.compute_centroids(data, full, k), ref_c, rtol=1e-5, atol=1e-4)
    ok("max_sim / compute_centroids == MultiKMeans.get_labels / compute_centroids")

    # residual tables (index/IVFPQIndex.py:160-170, 366-405) on a CPU-trained residual index
    np.random.seed(7)
    base = _nonneg(rng, 16, 1200)
    idx = tq.index.IVFPQIndex(d_vector=16, n_subvectors=4, n_cells=8, initial_size=256, device="cpu",
                              pq_use_residual=True)
    idx.train(torch.from_numpy(base.copy()))
    xq = _nonneg(rng, 16, 9)
    p1, p2 = idx.precomputed_adc_residual_precomputed(torch.from_numpy(xq.copy()))
    vq_cb, pq_cb = idx.vq_codec.codebook.numpy(), idx.pq_codec.codebook.numpy()
    np.testing.assert_allclose(orc.residual_part1(xq, pq_cb), np.ascontiguousarray(p1.numpy()), rtol=1e-4,
                               atol=1e-4 * np.abs(p1.numpy()).max())
    np.testing.assert_allclose(orc.residual_part2(vq_cb, pq_cb), np.ascontiguousarray(p2.numpy()), rtol=1e-4,
                               atol=1e-4 * np.abs(p2.numpy()).max())
    ok("residual_part1 / residual_part2 == IVFPQIndex.precomputed_adc_residual_precomputed")
    return done


if __name__ == "__main__":
    run()
