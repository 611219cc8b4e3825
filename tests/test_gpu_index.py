"""GPU: the drop-in IVFPQIndex (train / add / search / remove / state_dict) against the oracle."""
import io

import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import ivfpq_oracle as orc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _scan_layout_at_every_m(monkeypatch):
    """the index tests use small m: keep the scan layout (and its incremental scatter) exercised"""
    from torchpq_amd.index import IVFPQIndex
    monkeypatch.setattr(IVFPQIndex, "packed_min_subvectors", 0)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def N(t):
    return t.detach().cpu().numpy()


def _index_from_fixture(fx, **kw):
    from torchpq_amd.index import IVFPQIndex
    idx = IVFPQIndex(d_vector=int(fx["d"]), n_subvectors=int(fx["m"]), n_cells=int(fx["n_cells"]),
                     device=DEV, **kw)
    sd = {k[3:]: torch.from_numpy(v.copy()) for k, v in fx.items() if k.startswith("sd.")}
    idx.load_state_dict(sd)  # CPU tensors in, buffers land on the GPU
    return idx


def _expected_search(idx, x, k):
    """Oracle search driven by the index's own coarse step (cells / n_probe_list), with the LUT in
    the kernel's arithmetic (ascending-k fma chains == c_oracle.adc_lut): exact equality."""
    xq = T(x)
    _, cells, npl = idx.probe(xq)
    cells, npl = N(cells), N(npl)
    lut = c_oracle.adc_lut(x, N(idx.pq_codec.codebook), idx.distance)
    cs, sz = N(idx._cell_start)[cells], N(idx._cell_size)[cells]
    v, a = c_oracle.scan_topk(N(idx._storage), lut, N(idx._is_empty), cs, sz, npl, k)
    return v, orc.get_id_by_address(N(idx._address2id), a), cells, npl


@pytest.mark.parametrize("name", ["fx_tiny", "fx_m16"])
@pytest.mark.parametrize("smart", [False, True])
@pytest.mark.parametrize("packed", [False, True])
def test_search_on_reference_trained_index(name, smart, packed, request):
    fx = request.getfixturevalue(name)
    idx = _index_from_fixture(fx)
    assert idx.n_items == int(fx["n"]) and idx.max_id == int(fx["n"]) - 1
    idx.n_probe = int(fx["n_probe"])
    idx.use_smart_probing = smart
    idx.use_packed_layout = packed
    for k in fx["ks"]:
        k = int(k)
        v, i = idx.search(T(fx["queries"]), k=k)
        ev, ei, cells, npl = _expected_search(idx, fx["queries"], k)
        assert np.array_equal(N(v), ev)
        assert np.array_equal(N(i), ei)
        # against the reference-generated vectors: same cells, same ids, values within 1e-4
        assert np.array_equal(cells, fx["ref_cells"])
        if smart:
            assert np.array_equal(npl, fx["ref_nprobe_list"])
        gv, gi = fx[f"orc_vals_s{int(smart)}_k{k}"], fx[f"orc_ids_s{int(smart)}_k{k}"]
        fin = np.isfinite(gv)
        np.testing.assert_allclose(N(v)[fin], gv[fin], rtol=1e-4)
        assert (N(i) == gi).mean() > 0.995  # LUT rounding may swap exact near-ties


@pytest.mark.parametrize("name", ["fx_tiny", "fx_residual"])
def test_state_dict_keys_shapes_dtypes_match_reference(name, request):
    """state_dict interchange: after loading a reference-made state_dict the index
    saves exactly the reference's keys with the same shapes and dtypes, and the values round-trip."""
    fx = request.getfixturevalue(name)
    kw = {"pq_use_residual": True} if name == "fx_residual" else {}
    idx = _index_from_fixture(fx, **kw)
    ref = {k[3:]: v for k, v in fx.items() if k.startswith("sd.")}
    mine = {k: v for k, v in idx.state_dict().items() if v is not None}
    assert set(mine) == set(ref), set(mine) ^ set(ref)
    for k, v in ref.items():
        got = N(mine[k])
        assert got.shape == v.shape and got.dtype == v.dtype, (k, got.shape, v.shape, got.dtype, v.dtype)
        assert np.array_equal(got, v), k


def test_train_add_search_end_to_end_recall():
    from torchpq_amd.index import IVFPQIndex
    rng = np.random.default_rng(0)
    d, n, nq = 64, 20000, 200
    centers = rng.standard_normal((d, 50)) * 4
    base = (centers[:, rng.integers(0, 50, n)] + rng.standard_normal((d, n))).astype(np.float32)
    queries = (base[:, rng.choice(n, nq, replace=False)] + 0.05 * rng.standard_normal((d, nq))).astype(np.float32)
    np.random.seed(0)
    idx = IVFPQIndex(d_vector=d, n_subvectors=16, n_cells=64, initial_size=64, device=DEV)
    xb = T(base)
    before = xb.clone()
    idx.train(xb)
    assert torch.equal(xb, before), "train must not mutate its input"
    ids = idx.add(xb)
    assert torch.equal(ids, torch.arange(n, device=DEV))
    assert idx.n_items == n and idx.max_id == n - 1
    assert int(N(idx._cell_size).sum()) == n
    # inverted-list invariants
    st, sz, cap = N(idx._cell_start), N(idx._cell_size), N(idx._cell_capacity)
    assert np.array_equal(st, np.cumsum(cap) - cap) and np.all(sz <= cap)
    ie = N(idx._is_empty)
    for c in range(64):
        assert np.all(ie[st[c]:st[c] + sz[c]] == 0) and np.all(ie[st[c] + sz[c]:st[c] + cap[c]] == 1)
    # codes stored == encode(x) and each vector sits in its nearest cell
    adr = idx.get_address_by_id(ids)
    assert torch.equal(idx.get_data_by_address(adr), idx.encode(xb))
    assert torch.equal(idx.get_cell_by_address(adr), idx.vq_codec.encode(xb))
    _, cells_or = c_oracle.max_sim(base[None], N(idx.vq_codec.codebook)[None], "euclidean", "expanded")
    assert np.array_equal(N(idx.vq_codec.encode(xb)), cells_or[0])

    idx.n_probe = 16
    idx.use_smart_probing = False
    v, i = idx.search(T(queries), k=10)
    ev, ei, _, _ = _expected_search(idx, queries, 10)
    assert np.array_equal(N(v), ev) and np.array_equal(N(i), ei)
    # recall@10 of the true nearest neighbour (exact search on the raw vectors)
    gt = np.argmin(((queries.T[:, None, :] - base.T[None, :, :]) ** 2).sum(-1), axis=1)
    recall = np.mean([gt[q] in N(i)[q] for q in range(nq)])
    assert recall > 0.9, recall
    # returned values are -|q - decode(code)|^2 of the returned ids
    recon = N(idx.decode(idx.get_data_by_address(idx.get_address_by_id(i[:, 0].contiguous()))))
    exact = -((queries - recon) ** 2).sum(0)
    np.testing.assert_allclose(N(v)[:, 0], exact, rtol=1e-3, atol=1e-3)


def test_state_dict_roundtrip_and_growth():
    from torchpq_amd.index import IVFPQIndex
    rng = np.random.default_rng(1)
    d, n = 32, 5000
    base = np.abs(rng.standard_normal((d, n)) * 20).astype(np.float32)
    np.random.seed(1)
    idx = IVFPQIndex(d_vector=d, n_subvectors=8, n_cells=16, initial_size=8, expand_mode="double",
                     device=DEV)
    idx.train(T(base))
    custom = torch.arange(n, device=DEV) * 3 + 11
    idx.add(T(base[:, :3000]), ids=custom[:3000])      # forces many expansions
    idx.add(T(base[:, 3000:]), ids=custom[3000:])
    assert idx.n_items == n and idx.max_id == int(custom.max())
    idx.n_probe = 4
    q = T(base[:, :50] + 1.0)
    v1, i1 = idx.search(q, k=20)
    buf = io.BytesIO()
    torch.save(idx.state_dict(), buf)
    buf.seek(0)
    sd = torch.load(buf, map_location="cpu")
    assert set(sd) == {"_address2id", "_storage", "_cell_start", "_cell_size", "_cell_capacity",
                       "_is_empty", "vq_codec._is_trained", "vq_codec.kmeans.centroids",
                       "pq_codec._is_trained", "pq_codec.kmeans.centroids"}  # reference keys
    assert sd["_storage"].shape[0] == 2 and sd["_storage"].shape[2] == 4
    assert sd["vq_codec.kmeans.centroids"].shape == (d, 16)
    assert sd["pq_codec.kmeans.centroids"].shape == (8, 4, 256)
    idx2 = IVFPQIndex(d_vector=d, n_subvectors=8, n_cells=16, device=DEV)
    idx2.load_state_dict(sd)
    idx2.n_probe = 4
    v2, i2 = idx2.search(q, k=20)
    assert torch.equal(v1, v2) and torch.equal(i1, i2)
    assert idx2.max_id == idx.max_id and idx2.n_items == n
    # placement parity with the oracle container for the same codes / cells
    codes, cells = N(idx.encode(T(base))), N(idx.vq_codec.encode(T(base)))
    st = orc.ContainerState(8, 16, 8)
    st.add(codes[:, :3000], cells[:3000], N(custom[:3000]))
    st.add(codes[:, 3000:], cells[3000:], N(custom[3000:]))
    assert np.array_equal(N(idx._storage), st.storage)
    assert np.array_equal(N(idx._address2id), st.address2id)
    assert np.array_equal(N(idx._is_empty), st.is_empty)
    assert np.array_equal(N(idx._cell_start), st.cell_start)
    assert np.array_equal(N(idx._cell_capacity), st.cell_capacity)


def test_container_add_sequences_match_reference(fx_container):
    from torchpq_amd.container import CellContainer
    fx = fx_container
    for case in (0, 1):
        double, step = fx[f"c{case}_mode"]
        c = CellContainer(code_size=8, n_cells=5, dtype="uint8", device=DEV, initial_size=4,
                          expand_step_size=int(step), expand_mode="double" if double else "step",
                          use_inverse_id_mapping=True, contiguous_size=4)
        for b in range(4):
            ids_in = fx[f"c{case}_b{b}_ids_in"]
            ids, adr = c.add(T(fx[f"c{case}_b{b}_codes"]), T(fx[f"c{case}_b{b}_cells"]),
                             ids=T(ids_in) if ids_in.size else None, return_address=True)
            assert np.array_equal(N(ids), fx[f"c{case}_b{b}_ref_ids"])
            assert np.array_equal(N(adr), fx[f"c{case}_b{b}_ref_adr"])
            for k in ["_storage", "_cell_start", "_cell_size", "_cell_capacity", "_is_empty",
                      "_address2id"]:
                assert np.array_equal(N(getattr(c, k)), fx[f"c{case}_b{b}_sd{k}"]), (case, b, k)
            assert c.max_id == int(fx[f"c{case}_b{b}_max_id"])
        probe = T(fx[f"c{case}_probe_adr"])
        assert np.array_equal(N(c.get_cell_by_address(probe)), fx[f"c{case}_ref_cell_of_adr"])
        assert np.array_equal(N(c.get_id_by_address(probe)), fx[f"c{case}_ref_id_of_adr"])
        assert np.array_equal(N(c.get_data_by_address(probe)), fx[f"c{case}_ref_data_of_adr"])


def test_remove_keeps_cells_dense_and_results_exact(fx_m16):
    idx = _index_from_fixture(fx_m16)
    idx.n_probe = int(fx_m16["n_probe"])
    idx.use_smart_probing = False
    rng = np.random.default_rng(2)
    victims = torch.from_numpy(rng.choice(int(fx_m16["n"]), 700, replace=False)).to(DEV)
    n0 = idx.n_items
    idx.remove(ids=victims)
    assert idx.n_items == n0 - 700
    st, sz, cap, ie = N(idx._cell_start), N(idx._cell_size), N(idx._cell_capacity), N(idx._is_empty)
    a2i = N(idx._address2id)
    for c in range(st.size):
        assert np.all(ie[st[c]:st[c] + sz[c]] == 0) and np.all(ie[st[c] + sz[c]:st[c] + cap[c]] == 1)
        assert np.all(a2i[st[c] + sz[c]:st[c] + cap[c]] == -1)
    assert np.all(N(idx.get_address_by_id(victims)) == -1)
    left = np.setdiff1d(np.arange(int(fx_m16["n"])), N(victims))
    assert np.array_equal(np.sort(a2i[a2i >= 0]), left)
    v, i = idx.search(T(fx_m16["queries"]), k=10)
    ev, ei, _, _ = _expected_search(idx, fx_m16["queries"], 10)
    assert np.array_equal(N(v), ev) and np.array_equal(N(i), ei)
    assert not np.isin(N(i), N(victims)).any()
    # removed ids can be added back and found again
    back = T(fx_m16["base"][:, N(victims)[:50]])
    idx.add(back, ids=victims[:50])
    assert np.all(N(idx.get_address_by_id(victims[:50])) >= 0)


def test_api_asserts_and_cosine():
    from torchpq_amd.index import IVFPQIndex
    with pytest.raises(AssertionError):
        IVFPQIndex(d_vector=30, n_subvectors=8, device=DEV)  # d % m != 0
    rng = np.random.default_rng(3)
    base = rng.standard_normal((32, 3000)).astype(np.float32)
    np.random.seed(3)
    idx = IVFPQIndex(d_vector=32, n_subvectors=8, n_cells=8, distance="cosine", device=DEV)
    with pytest.raises(AssertionError):
        idx.search(T(base[:, :4]), k=1)  # not trained
    idx.train(T(base))
    idx.add(T(base))
    with pytest.raises(AssertionError):
        idx.search(T(base[:31, :4]), k=1)
    with pytest.raises(AssertionError):
        idx.search(T(base[:, :4]), k=2000)
    with pytest.raises(AssertionError):
        idx.vq_codec_max_iter = 3  # already trained
    idx.n_probe = 8
    v, i = idx.search(T(base[:, :100]), k=5)
    assert N(v).max() <= 1.0 + 1e-3 and (N(i)[:, 0] == np.arange(100)).mean() > 0.8
    xn = base[:, :100] / (np.linalg.norm(base[:, :100], axis=0, keepdims=True) + 1e-9)
    ev, ei, _, _ = _expected_search(idx, xn.astype(np.float32), 5)
    np.testing.assert_allclose(N(v), ev, rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("use_precomputed", [True, False])
def test_residual_index_end_to_end(use_precomputed):
    """pq_use_residual=True: train / add / encode / decode / search against the oracle."""
    from torchpq_amd.index import IVFPQIndex
    rng = np.random.default_rng(6)
    d, n, nq, k = 32, 6000, 40, 10
    centers = rng.standard_normal((d, 30)) * 4
    base = (centers[:, rng.integers(0, 30, n)] + rng.standard_normal((d, n))).astype(np.float32)
    queries = (base[:, :nq] + 0.05 * rng.standard_normal((d, nq))).astype(np.float32)
    np.random.seed(6)
    idx = IVFPQIndex(d_vector=d, n_subvectors=8, n_cells=16, initial_size=512, device=DEV,
                     pq_use_residual=True)
    assert idx.use_precomputed
    xb = T(base)
    keep = xb.clone()
    idx.train(xb)
    assert torch.equal(xb, keep)
    idx.add(xb)
    idx.use_precomputed = use_precomputed
    pq_code, vq_code = idx.encode(xb)
    adr = idx.get_address_by_id(torch.arange(n, device=DEV))
    assert torch.equal(idx.get_data_by_address(adr), pq_code)
    assert torch.equal(idx.get_cell_by_address(adr), vq_code)
    recon = idx.decode((pq_code, vq_code))
    err_res = float(((recon - xb) ** 2).sum(0).mean())
    err_vq = float(((idx.vq_codec.decode(vq_code) - xb) ** 2).sum(0).mean())
    assert err_res < 0.7 * err_vq  # the residual codes refine the coarse reconstruction
    idx.n_probe = 6
    idx.use_smart_probing = False
    v, i = idx.search(T(queries), k=k)
    assert (N(i)[:, 0] == np.arange(nq)).mean() > 0.9
    # oracle: same cells / base sims, tables in the kernel's arithmetic
    topk_sims, cells, npl = idx.probe(T(queries))
    cells_n, npl_n = N(cells), N(npl)
    cs, sz = N(idx._cell_start)[cells_n], N(idx._cell_size)[cells_n]
    if use_precomputed:
        p1, p2 = idx.precomputed_adc_residual_precomputed(T(queries))
        ev, ea = c_oracle.scan_topk_residual(N(idx._storage), N(p1), N(p2.contiguous()), cells_n,
                                             N(topk_sims), N(idx._is_empty), cs, sz, npl_n, k)
    else:
        full = idx.precomputed_adc_residual(T(queries), cells)
        ev, ea = c_oracle.scan_topk_residual(N(idx._storage), None, None, None, N(topk_sims),
                                             N(idx._is_empty), cs, sz, npl_n, k, full=N(full))
    assert np.array_equal(N(v), ev)
    assert np.array_equal(N(i), orc.get_id_by_address(N(idx._address2id), ea))
    # values are -|q - (centroid + decoded residual)|^2 of the returned ids
    top = i[:, 0].contiguous()
    rec = N(idx.decode((pq_code[:, top], vq_code[top])))
    exact = -((queries - rec) ** 2).sum(0)
    np.testing.assert_allclose(N(v)[:, 0], exact, rtol=2e-3, atol=2e-3 * np.abs(exact).max())
    if use_precomputed:
        # scan layout (default) == reference layout == table-fed scan layout, bit for bit; the
        # per-slot constants of the scan layout follow the codes through add / remove
        def three_ways():
            out = []
            for packed, fused in ((True, True), (False, True), (True, False)):
                idx.use_packed_layout, idx.use_fused_lut = packed, fused
                out.append(idx.search(T(queries), k=k))
            idx.use_packed_layout = idx.use_fused_lut = True
            return out
        for vv, ii in three_ways():
            assert torch.equal(vv, v) and torch.equal(ii, i)
        idx.add(T((base[:, :500] + 0.5).astype(np.float32)), ids=torch.arange(n, n + 500, device=DEV))
        idx.remove(ids=torch.arange(0, 200, device=DEV))
        (va, ia), (vb, ib), (vc, ic) = three_ways()
        assert not torch.equal(ia, i)
        assert torch.equal(va, vb) and torch.equal(ia, ib) and torch.equal(va, vc) and torch.equal(ia, ic)


@pytest.mark.parametrize("name", ["fx_c1", "fx_ties", "fx_tomb"])
def test_search_on_reference_built_state_dicts(name, request):
    """load_state_dict of reference-built indexes (C1 shape; duplicates; tombstones inside cells)
    and search: same ids as the oracle driven by the index's own coarse step."""
    fx = request.getfixturevalue(name)
    idx = _index_from_fixture(fx)
    idx.n_probe = int(fx["n_probe"])
    idx.use_smart_probing = False
    if name == "fx_tomb":
        assert idx._has_holes
    for k in (1, 10):
        v, i = idx.search(T(fx["queries"]), k=k)
        ev, ei, cells, npl = _expected_search(idx, fx["queries"], k)
        assert np.array_equal(N(v), ev) and np.array_equal(N(i), ei)
        if name == "fx_tomb":
            assert not np.isin(N(i), fx["dead_ids"]).any()
    # the coarse step picks the reference's cells (as a set; order differs only inside fp32 ties)
    _, cells, _ = idx.probe(T(fx["queries"]))
    same = [set(a) == set(b) for a, b in zip(N(cells).tolist(), fx["ref_cells"].tolist())]
    assert np.mean(same) > 0.9


@pytest.mark.parametrize("residual", [False, True])
def test_graphed_search_replays_search(residual):
    """search() is sync-free and captures into one HIP graph; replays equal eager calls."""
    from torchpq_amd.index import IVFPQIndex
    rng = np.random.default_rng(12)
    d, n, nq, k = 32, 4000, 9, 10
    base = rng.standard_normal((d, n)).astype(np.float32)
    np.random.seed(12)
    idx = IVFPQIndex(d_vector=d, n_subvectors=8, n_cells=16, initial_size=512, device=DEV,
                     pq_use_residual=residual)
    idx.train(T(base))
    idx.add(T(base))
    idx.n_probe = 5
    g = idx.graphed_search(nq, k=k)
    for seed in (1, 2, 3):
        q = T(np.random.default_rng(seed).standard_normal((d, nq)).astype(np.float32))
        v, i = g(q)
        ev, ei = idx.search(q, k=k)
        assert torch.equal(v, ev) and torch.equal(i, ei)
    # a stale graph refuses to replay: knob change, re-train (new codebook buffers), residual
    # table rebuild, add -- each detected (only _codes_version used to be checked)
    idx.n_probe = 3
    with pytest.raises(RuntimeError, match="n_probe"):
        g(q)
    idx.n_probe = 5
    assert g.stale_reason() is None
    g(q)
    idx.use_smart_probing = False
    with pytest.raises(RuntimeError, match="use_smart_probing"):
        g(q)
    idx.use_smart_probing = True
    if residual:
        idx.use_precomputed = True   # rebuilds the part2 table: new buffer
        with pytest.raises(RuntimeError, match="part2|slot_term|cell_bound"):
            g(q)
        g = idx.graphed_search(nq, k=k)
        v, i = g(q)
        ev, ei = idx.search(q, k=k)
        assert torch.equal(v, ev) and torch.equal(i, ei)
    np.random.seed(13)
    idx.train(T(base), force_retrain=True)  # registers new codebooks; codes are now meaningless
    with pytest.raises(RuntimeError, match="codebook|part2"):
        g(q)
    g = idx.graphed_search(nq, k=k)
    g(q)
    idx.add(T(base[:, :10] + 1), ids=torch.arange(n, n + 10, device=DEV))
    with pytest.raises(RuntimeError, match="_codes_version"):
        g(q)


def test_flat_index_exact_search():
    """FlatIndex: exact neighbours, ids, removal; doubles as recall ground truth."""
    from torchpq_amd.index import FlatIndex, IVFPQIndex
    rng = np.random.default_rng(9)
    d, n, nq, k = 24, 5000, 64, 20
    base = rng.standard_normal((d, n)).astype(np.float32)
    queries = rng.standard_normal((d, nq)).astype(np.float32)
    flat = FlatIndex(d_vector=d, initial_size=16, device=DEV)
    ids = torch.arange(n, device=DEV) * 2 + 5
    flat.add(T(base[:, :3000]), ids=ids[:3000])
    flat.add(T(base[:, 3000:]), ids=ids[3000:])
    assert flat.n_items == n
    v, i = flat.search(T(queries), k=k)
    d2 = -((queries.T[:, None, :] - base.T[None, :, :]) ** 2).sum(-1)
    order = np.argsort(-d2, axis=1, kind="stable")[:, :k]
    assert (N(i) == N(ids)[order]).mean() > 0.999  # GEMM-form rounding may swap exact near-ties
    np.testing.assert_allclose(N(v), np.take_along_axis(d2, order, 1), rtol=1e-4, atol=1e-4)
    flat.remove(ids=ids[order[:, 0]].unique())
    v2, i2 = flat.search(T(queries), k=1)
    assert not np.isin(N(i2)[:, 0], N(ids)[order[:, 0]]).any()
    # recall of the IVFPQ index against it
    np.random.seed(9)
    ivf = IVFPQIndex(d_vector=d, n_subvectors=12, n_cells=32, initial_size=256, device=DEV)
    ivf.train(T(base))
    ivf.add(T(base), ids=ids)
    ivf.n_probe = 32
    flat2 = FlatIndex(d_vector=d, device=DEV)
    flat2.add(T(base), ids=ids)
    _, gt = flat2.search(T(queries), k=1)
    _, got = ivf.search(T(queries), k=k)
    recall = (got == gt).any(dim=1).float().mean().item()
    assert recall > 0.8, recall


def test_search_edge_cases_many_probes_batches_empty():
    """n_probe = 1 and > 64 (probe table built in 64-probe chunks), query batches split by
    max_query_batch, search on an empty / untouched cell set."""
    from torchpq_amd.index import IVFPQIndex
    rng = np.random.default_rng(12)
    d, n, nq = 32, 30000, 150
    base = (rng.standard_normal((d, n)) * 3).astype(np.float32)
    queries = (rng.standard_normal((d, nq)) * 3).astype(np.float32)
    np.random.seed(12)
    idx = IVFPQIndex(d_vector=d, n_subvectors=8, n_cells=160, initial_size=32, device=DEV)
    idx.train(T(base[:, :8000]))
    idx.use_smart_probing = False
    idx.n_probe = 5
    v, i = idx.search(T(queries), k=7)  # nothing added yet
    assert torch.isneginf(v).all() and (i == -1).all()
    idx.add(T(base))
    for n_probe, k in [(1, 1), (1, 20), (100, 10), (160, 64)]:
        idx.n_probe = n_probe
        v, i = idx.search(T(queries), k=k)
        ev, ei, _, _ = _expected_search(idx, queries, k)
        assert np.array_equal(N(v), ev) and np.array_equal(N(i), ei), (n_probe, k)
    idx.n_probe = 12
    v_all, i_all = idx.search(T(queries), k=10)
    idx.max_query_batch = 64  # 150 queries -> 3 batches
    v_b, i_b = idx.search(T(queries), k=10)
    assert torch.equal(v_all, v_b) and torch.equal(i_all, i_b)
    idx.use_smart_probing = True
    idx.max_query_batch = 32768
    v_s, i_s = idx.search(T(queries), k=10)
    ev, ei, _, npl = _expected_search(idx, queries, 10)
    assert np.array_equal(N(v_s), ev) and np.array_equal(N(i_s), ei)
    assert npl.min() >= 1 and npl.max() <= 12


def test_remove_on_a_state_dict_with_holes(fx_tomb):
    """remove() on a foreign index whose cells hold tombstones INSIDE [start, start+size)
    used to trip a bare assert; now the cells are compacted first.  Result = oracle search on the
    surviving items; codes and ids of survivors are untouched."""
    fx = fx_tomb
    idx = _index_from_fixture(fx)
    assert idx._has_holes
    idx.n_probe = int(fx["n_probe"])
    idx.use_smart_probing = False
    live_before = N(idx._address2id)
    live_ids = np.sort(live_before[live_before >= 0])
    codes_before = {int(i): N(idx.get_data_by_address(idx.get_address_by_id(torch.tensor([i], device=DEV))))[:, 0]
                    for i in live_ids[:40]}
    victims = live_ids[::7]
    idx.remove(ids=T(victims))
    assert not idx._has_holes
    st, sz, cap = N(idx._cell_start), N(idx._cell_size), N(idx._cell_capacity)
    ie, a2i = N(idx._is_empty), N(idx._address2id)
    for c in range(idx.n_cells):  # dense cells again
        assert np.all(ie[st[c]:st[c] + sz[c]] == 0) and np.all(ie[st[c] + sz[c]:st[c] + cap[c]] == 1)
        assert np.all(a2i[st[c]:st[c] + sz[c]] >= 0) and np.all(a2i[st[c] + sz[c]:st[c] + cap[c]] == -1)
    left = np.sort(a2i[a2i >= 0])
    assert np.array_equal(left, np.setdiff1d(live_ids, victims))
    assert idx.n_items == left.shape[0]
    for i, code in codes_before.items():
        if i in set(victims.tolist()):
            continue
        adr = idx.get_address_by_id(torch.tensor([i], device=DEV))
        assert np.array_equal(N(idx.get_data_by_address(adr))[:, 0], code)
    v, i = idx.search(T(fx["queries"]), k=10)
    ev, ei, _, _ = _expected_search(idx, fx["queries"], 10)
    assert np.array_equal(N(v), ev) and np.array_equal(N(i), ei)
    assert not np.isin(N(i), victims).any() and not np.isin(N(i), fx["dead_ids"]).any()


def test_is_trained_sees_in_place_writes():
    """the cached trained flag follows in-place updates of the buffer"""
    from torchpq_amd.codec import VQCodec
    c = VQCodec(n_clusters=4).to(DEV)
    assert c.is_trained is False
    c._is_trained.fill_(True)
    assert c.is_trained is True
    c._is_trained.data = torch.tensor(False, device=DEV)
    assert c.is_trained is False
    c._is_trained.copy_(torch.tensor(True))
    assert c.is_trained is True


@pytest.mark.parametrize("packed", [False, True])
def test_cosine_index_on_reference_trained_state_dict(fx_cosine, packed):
    """distance="cosine" end to end on a reference-trained index (fx_cosine): search() normalises
    the queries (IVFPQIndex.py:480-481), probes with the euclidean coarse quantiser, builds the
    dot-product LUT; results bit-equal the oracle on the index's own probe, the LUT kernel matches
    the reference's precompute_adc, ids match the golden oracle run."""
    import torchpq_amd.kernels as K
    fx = fx_cosine
    idx = _index_from_fixture(fx, distance="cosine")
    idx.n_probe = int(fx["n_probe"])
    idx.use_smart_probing = False
    idx.use_packed_layout = packed
    lut = K.AdcLutHip()(T(fx["queries_normalized"]), idx.pq_codec.codebook, "cosine")
    np.testing.assert_allclose(N(lut), fx["ref_lut"], rtol=1e-4, atol=1e-6)
    assert np.array_equal(N(lut), c_oracle.adc_lut(fx["queries_normalized"], N(idx.pq_codec.codebook), "cosine"))
    xn = N(util_normalize(T(fx["queries"])))
    for k in (1, 10):
        v, i = idx.search(T(fx["queries"]), k=k)           # un-normalised in, as a user would call it
        ev, ei, cells, _ = _expected_search(idx, N(util_normalize(T(fx["queries"]))), k)
        assert np.array_equal(N(v), ev) and np.array_equal(N(i), ei)
        same_cells = [set(a) == set(b) for a, b in zip(cells.tolist(), fx["ref_cells"].tolist())]
        assert np.mean(same_cells) > 0.9
        fin = np.isfinite(fx[f"orc_vals_k{k}"])
        np.testing.assert_allclose(N(v)[fin], fx[f"orc_vals_k{k}"][fin], rtol=1e-3, atol=1e-5)
        assert (N(i) == fx[f"orc_ids_k{k}"]).mean() > 0.95
    assert np.allclose(xn, fx["queries_normalized"], atol=1e-6)
    # values are cosine similarities of the PQ reconstruction: within [-1, 1] up to PQ error
    assert float(v.max()) <= 1.05 and float(v.min()) >= -1.05


def util_normalize(x):
    from torchpq_amd import util
    return util.normalize(x, dim=0)


def test_add_uses_the_fast_coarse_assign_and_cells_stay_exact(monkeypatch):
    """IVFPQIndex.add -> VQCodec.encode -> KMeans.predict runs tpq_coarse_assign from
    KMeans.fast_predict_min_work on (lowered here so a small index takes it): cells equal the
    oracle's fp32 arg-max, bit for bit, and equal what the fp32 kernel path produces."""
    from torchpq_amd import kernels as K
    from torchpq_amd.clustering import KMeans
    from torchpq_amd.index import IVFPQIndex
    calls = []
    orig = K.CoarseAssignHip.__call__
    monkeypatch.setattr(K.CoarseAssignHip, "__call__",
                        lambda self, A, B, **kw: (calls.append(A.shape), orig(self, A, B, **kw))[1])
    rng = np.random.default_rng(11)
    d, n = 64, 30000
    centers = rng.standard_normal((d, 300)) * 4
    base = (centers[:, rng.integers(0, 300, n)] + rng.standard_normal((d, n))).astype(np.float32)
    np.random.seed(1)
    idx = IVFPQIndex(d_vector=d, n_subvectors=16, n_cells=256, initial_size=64, device=DEV)
    idx.train(T(base))
    calls.clear()  # (the coarse quantiser's training takes its labels from the same kernel)
    monkeypatch.setattr(KMeans, "fast_predict_min_work", 1 << 60)
    slow = N(idx.vq_codec.encode(T(base)))
    assert not calls
    monkeypatch.setattr(KMeans, "fast_predict_min_work", 0)
    ids, adr = idx.add(T(base), return_address=True)
    assert calls and calls[0] == (d, n)
    cells = N(idx.get_cell_by_address(adr))
    _, cells_or = c_oracle.max_sim(base[None], N(idx.vq_codec.codebook)[None], "euclidean", "expanded")
    assert np.array_equal(cells, cells_or[0]) and np.array_equal(cells, slow)


def test_pq_encode_of_wide_subvectors_uses_the_selection_kernel_and_stays_exact(monkeypatch):
    """d_subvector = 16 (e.g. 768-d embeddings at m = 48): PQCodec.encode takes tpq_max_sim_select and
    returns the oracle's codes, bit for bit -- the same bytes the fp32 kernel path produces"""
    from torchpq_amd import kernels as K
    from torchpq_amd.codec import PQCodec
    rng = np.random.default_rng(5)
    d, m, n = 128, 8, 20000
    x = (rng.standard_normal((d, n)) * 3).astype(np.float32)
    np.random.seed(3)
    pq = PQCodec(d_vector=d, n_subvectors=m, n_clusters=256)
    pq.kmeans.max_iter = 3
    pq.train(T(x))
    calls = []
    orig = K.MaxSimSelectHip.__call__
    monkeypatch.setattr(K.MaxSimSelectHip, "__call__", lambda self, A, B: (calls.append(A.shape), orig(self, A, B))[1])
    monkeypatch.setattr(PQCodec, "select_min_work", 1 << 60)
    slow = N(pq.encode(T(x)))
    assert not calls
    monkeypatch.setattr(PQCodec, "select_min_work", 0)
    fast = N(pq.encode(T(x)))
    assert calls == [(m, d // m, n)]
    cb = N(pq.codebook)
    _, want = c_oracle.max_sim(x.reshape(m, d // m, n), cb, "euclidean", "expanded")
    assert fast.dtype == np.uint8 and np.array_equal(fast, want.astype(np.uint8)) and np.array_equal(fast, slow)


@pytest.mark.parametrize("distance", ["inner", "cosine"])
def test_fast_predict_for_inner_and_cosine_equals_the_fp32_path(distance, monkeypatch):
    """KMeans.predict through tpq_coarse_assign for the other two metrics: cosine normalises exactly as
    get_labels does and then takes the inner-product path; labels equal the fp32 kernel's"""
    from torchpq_amd.clustering import KMeans
    rng = np.random.default_rng(8)
    d, n, k = 48, 9000, 200
    x = T((rng.standard_normal((d, n)) * 2 + 0.5).astype(np.float32))
    km = KMeans(n_clusters=k, max_iter=3, distance=distance)
    np.random.seed(2)
    km.fit(x)
    monkeypatch.setattr(KMeans, "fast_predict_min_work", 1 << 60)
    slow = km.predict(x)
    monkeypatch.setattr(KMeans, "fast_predict_min_work", 0)
    fast = km.predict(x)
    assert torch.equal(fast, slow)
