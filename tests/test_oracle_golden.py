"""CPU: the oracle (numpy + C restatement) against the reference-generated golden vectors."""
import numpy as np
import pytest

from oracle import c_oracle
from oracle import ivfpq_oracle as orc

RTOL = 1e-4  # BASELINE.json north_star: float distances within 1e-4 relative


def _sd(fx, key):
    return fx["sd." + key]


@pytest.mark.parametrize("name", ["fx_tiny", "fx_m16", "fx_c1"])
def test_coarse_sims_and_lut_match_reference(name, request):
    fx = request.getfixturevalue(name)
    vq = _sd(fx, "vq_codec.kmeans.centroids")
    pq = _sd(fx, "pq_codec.kmeans.centroids")
    sims = orc.neg_sq_l2(fx["queries"], vq)
    np.testing.assert_allclose(sims, fx["ref_sims"], rtol=RTOL, atol=0)
    lut = orc.adc_lut(fx["queries"], pq)
    scale = np.abs(fx["ref_lut"]).max()
    np.testing.assert_allclose(lut, fx["ref_lut"], rtol=RTOL, atol=1e-6 * scale)
    lut_c = c_oracle.adc_lut(fx["queries"], pq)
    np.testing.assert_allclose(lut_c, fx["ref_lut"], rtol=RTOL, atol=1e-6 * scale)


@pytest.mark.parametrize("name", ["fx_tiny", "fx_m16", "fx_c1"])
def test_coarse_select_and_smart_probing(name, request):
    fx = request.getfixturevalue(name)
    v, c = orc.topk_desc(fx["ref_sims"], int(fx["n_probe"]))
    assert np.array_equal(v, fx["ref_topk_sims"])
    # torch.topk and the oracle agree on ids wherever values are distinct
    distinct = np.ones_like(v, dtype=bool)
    distinct[:, 1:] &= v[:, 1:] != v[:, :-1]
    distinct[:, :-1] &= v[:, 1:] != v[:, :-1]
    assert np.array_equal(c[distinct], fx["ref_cells"][distinct])
    npl = orc.smart_probing(fx["ref_topk_sims"], int(fx["n_probe"]))
    assert np.array_equal(npl, fx["ref_nprobe_list"])


@pytest.mark.parametrize("name", ["fx_tiny", "fx_m16"])
def test_encode_matches_reference_codes(name, request):
    fx = request.getfixturevalue(name)
    pq = _sd(fx, "pq_codec.kmeans.centroids")
    vq = _sd(fx, "vq_codec.kmeans.centroids")
    m, ds, _ = pq.shape
    base = fx["base"]
    n = base.shape[1]
    # reference codes as stored (gather back through _address2id)
    a2i = _sd(fx, "_address2id")
    adr = np.nonzero(a2i >= 0)[0]
    ids = a2i[adr]
    ref_codes = np.zeros((m, n), np.uint8)
    ref_codes[:, ids] = orc.storage_to_codes(_sd(fx, "_storage"), adr)
    _, labels = c_oracle.max_sim(base.reshape(m, ds, n), pq, "euclidean", "expanded")
    # the reference CPU path never labels the last point (MultiKMeans.py:352-353)
    agree = (labels[:, :-1].astype(np.uint8) == ref_codes[:, :-1]).mean()
    assert agree > 0.999, agree
    # where they differ the two candidates must be a near-tie
    _, cells = c_oracle.max_sim(base[None], vq[None], "euclidean", "expanded")
    ref_cell = orc.get_cell_by_address(adr, _sd(fx, "_cell_start"), _sd(fx, "_cell_capacity"))
    cell_of_id = np.empty(n, np.int64)
    cell_of_id[ids] = ref_cell
    assert (cells[0] == cell_of_id).mean() > 0.999


def test_decode_matches_reference(fx_tiny):
    pq = _sd(fx_tiny, "pq_codec.kmeans.centroids")
    rec = orc.pq_decode(pq, fx_tiny["ref_decode_codes"])
    assert np.array_equal(rec, fx_tiny["ref_decode"])


@pytest.mark.parametrize("name", ["fx_tiny", "fx_m16"])
def test_scan_value_identity_against_reference_decode(name, request):
    """sum_j LUT[j,q,code_j] == -|q - decode(code)|^2 with both sides from the reference."""
    fx = request.getfixturevalue(name)
    storage = _sd(fx, "_storage")
    cap = storage.shape[1]
    slots = np.arange(cap)
    for q in range(0, int(fx["nq"]), 3):
        v = orc.scan_values(storage, fx["ref_lut"][:, q, :], slots)
        ref = fx["ref_adc_exact"][q]
        np.testing.assert_allclose(v, ref, rtol=1e-4, atol=1e-4 * np.abs(ref).max())


@pytest.mark.parametrize("name", ["fx_tiny", "fx_m16"])
def test_scan_topk_numpy_vs_c_vs_golden(name, request):
    fx = request.getfixturevalue(name)
    storage, is_empty = _sd(fx, "_storage"), _sd(fx, "_is_empty")
    cs = _sd(fx, "_cell_start")[fx["ref_cells"]]
    sz = _sd(fx, "_cell_size")[fx["ref_cells"]]
    nq = int(fx["nq"])
    for smart in (0, 1):
        npl = fx["ref_nprobe_list"] if smart else np.full(nq, int(fx["n_probe"]), np.int64)
        for k in fx["ks"]:
            k = int(k)
            v, a = c_oracle.scan_topk(storage, fx["ref_lut"], is_empty, cs, sz, npl, k)
            assert np.array_equal(v, fx[f"orc_vals_s{smart}_k{k}"])
            assert np.array_equal(a, fx[f"orc_addr_s{smart}_k{k}"])
            ids = orc.get_id_by_address(_sd(fx, "_address2id"), a)
            assert np.array_equal(ids, fx[f"orc_ids_s{smart}_k{k}"])
            # brute-force check against the reference decode identity: the top-k values are the
            # k largest exact ADC values among the probed, non-empty slots
            for q in range(nq):
                slots = orc.probed_slots(cs[q], sz[q], npl[q])
                slots = slots[is_empty[slots] == 0]
                exact = np.sort(fx["ref_adc_exact"][q][slots])[::-1][:k]
                got = v[q][: exact.size]
                np.testing.assert_allclose(got, exact, rtol=1e-4, atol=1e-4 * np.abs(exact).max())
                assert np.all(a[q][exact.size:] == -1)


def test_container_add_sequences_bit_exact(fx_container):
    fx = fx_container
    for case in (0, 1):
        double, step = fx[f"c{case}_mode"]
        st = orc.ContainerState(8, 5, 4, expand_step_size=int(step),
                                expand_mode="double" if double else "step")
        for b in range(4):
            ids_in = fx[f"c{case}_b{b}_ids_in"]
            ids, adr = st.add(fx[f"c{case}_b{b}_codes"], fx[f"c{case}_b{b}_cells"],
                              ids_in if ids_in.size else None)
            assert np.array_equal(orc.get_ioa(fx[f"c{case}_b{b}_cells"]), fx[f"c{case}_b{b}_ref_ioa"])
            assert np.array_equal(ids, fx[f"c{case}_b{b}_ref_ids"])
            assert np.array_equal(adr, fx[f"c{case}_b{b}_ref_adr"])
            for k, mine in [("_storage", st.storage), ("_cell_start", st.cell_start),
                            ("_cell_size", st.cell_size), ("_cell_capacity", st.cell_capacity),
                            ("_is_empty", st.is_empty), ("_address2id", st.address2id)]:
                assert np.array_equal(mine, fx[f"c{case}_b{b}_sd{k}"]), (case, b, k)
            assert st.max_id == int(fx[f"c{case}_b{b}_max_id"])
        probe = fx[f"c{case}_probe_adr"]
        assert np.array_equal(orc.get_cell_by_address(probe, st.cell_start, st.cell_capacity),
                              fx[f"c{case}_ref_cell_of_adr"])
        assert np.array_equal(orc.get_id_by_address(st.address2id, probe), fx[f"c{case}_ref_id_of_adr"])
        assert np.array_equal(orc.storage_to_codes(st.storage, probe), fx[f"c{case}_ref_data_of_adr"])


def test_kmeans_assign_update_match_reference(fx_kmeans):
    fx = fx_kmeans
    data, init = fx["data"], fx["init"]
    for numerics in ("direct", "expanded"):
        vals, labels = c_oracle.max_sim(data, init, "euclidean", numerics)
        same = labels[:, :-1] == fx["ref_labels"]
        # disagreements only at near-ties of the two best centroids
        gap = fx["ref_sims_top2gap"][:, :-1]
        assert same.mean() > 0.999
        assert np.all(gap[~same] <= 1e-3 * np.abs(fx["ref_maxsims"][~same]) + 1e-3)
        np.testing.assert_allclose(vals[:, :-1][same], fx["ref_maxsims"][same], rtol=1e-4,
                                   atol=1e-2)
    cen = orc.compute_centroids(data, fx["labels_for_update"], init.shape[2])
    np.testing.assert_allclose(cen, fx["ref_centroids"], rtol=1e-5, atol=1e-4)
    v_np, l_np = orc.max_sim(data[:1, :, :512], init[:1], "euclidean", "direct")
    v_c, l_c = c_oracle.max_sim(data[:1, :, :512], init[:1], "euclidean", "direct")
    assert np.array_equal(l_np, l_c) and np.array_equal(v_np, v_c)


def test_residual_tables_and_scan_against_reference(fx_residual):
    """pq_use_residual=True: part1/part2 vs the reference's tables, the residual scan identity
    base + sum_j (part1 + part2)[j, code_j] == -|q - (centroid + decode(code))|^2 through reference
    functions, numpy vs C oracle, mode A (part1 + part2) vs mode B (full LUT)."""
    fx = fx_residual
    pq = _sd(fx, "pq_codec.kmeans.centroids")
    vq = _sd(fx, "vq_codec.kmeans.centroids")
    p1 = orc.residual_part1(fx["queries"], pq)
    np.testing.assert_allclose(p1, fx["ref_part1"], rtol=1e-4, atol=1e-5 * np.abs(fx["ref_part1"]).max())
    p2 = orc.residual_part2(vq, pq)
    np.testing.assert_allclose(p2, fx["ref_part2"], rtol=1e-4, atol=1e-5 * np.abs(fx["ref_part2"]).max())
    # the per-(query, probe) table equals part1 + part2[cell] up to fp32 rounding
    cells = fx["ref_cells"]
    comb = fx["ref_part1"][:, None] + fx["ref_part2"][cells]
    np.testing.assert_allclose(comb, fx["ref_full"], rtol=1e-4, atol=1e-4 * np.abs(comb).max())
    storage, is_empty = _sd(fx, "_storage"), _sd(fx, "_is_empty")
    cs, sz = _sd(fx, "_cell_start")[cells], _sd(fx, "_cell_size")[cells]
    nq, n_probe = cells.shape
    npl = np.full(nq, n_probe, np.int64)
    for k in (1, 10, 100):
        v, a = c_oracle.scan_topk_residual(storage, fx["ref_part1"], fx["ref_part2"], cells,
                                           fx["ref_topk_sims"], is_empty, cs, sz, npl, k)
        assert np.array_equal(v, fx[f"orc_vals_k{k}"]) and np.array_equal(a, fx[f"orc_addr_k{k}"])
        v2, a2 = c_oracle.scan_topk_residual(storage, None, None, None, fx["ref_topk_sims"],
                                             is_empty, cs, sz, npl, k, full=fx["ref_full"])
        assert np.array_equal(v2, fx[f"orc_full_vals_k{k}"]) and np.array_equal(a2, fx[f"orc_full_addr_k{k}"])
        for q in range(nq):
            slots = orc.probed_slots(cs[q], sz[q], npl[q])
            slots = slots[is_empty[slots] == 0]
            exact = np.sort(fx["ref_adc_exact"][q][slots])[::-1][:k]
            np.testing.assert_allclose(v[q][:exact.size], exact, rtol=1e-4,
                                       atol=2e-4 * np.abs(fx["ref_adc_exact"][q]).max())


def test_c1_shape_scan_numpy_vs_c_vs_golden(fx_c1):
    """BASELINE configs[0] shape (d=128, m=16, n_cells=256, nprobe=8, k=10) on a reference-trained
    index: the C and numpy restatements agree with the committed vectors."""
    fx = fx_c1
    storage, is_empty = _sd(fx, "_storage"), _sd(fx, "_is_empty")
    cs = _sd(fx, "_cell_start")[fx["ref_cells"]]
    sz = _sd(fx, "_cell_size")[fx["ref_cells"]]
    nq = int(fx["nq"])
    for smart in (0, 1):
        npl = fx["ref_nprobe_list"] if smart else np.full(nq, int(fx["n_probe"]), np.int64)
        v, a = c_oracle.scan_topk(storage, fx["ref_lut"], is_empty, cs, sz, npl, 10)
        assert np.array_equal(v, fx[f"orc_vals_s{smart}_k10"])
        assert np.array_equal(a, fx[f"orc_addr_s{smart}_k10"])
        v2, a2 = orc.scan_topk(storage, fx["ref_lut"], is_empty, cs[:6], sz[:6], npl[:6], 10)
        assert np.array_equal(v2, v[:6]) and np.array_equal(a2, a[:6])


def test_tie_policy_on_reference_duplicates(fx_ties):
    """The reference added 400 vectors twice: both copies share a cell and a code, so their values
    tie exactly; the restatement orders ties by ascending address and returns both ids."""
    fx = fx_ties
    storage, is_empty = _sd(fx, "_storage"), _sd(fx, "_is_empty")
    cs = _sd(fx, "_cell_start")[fx["ref_cells"]]
    sz = _sd(fx, "_cell_size")[fx["ref_cells"]]
    npl = np.full(int(fx["nq"]), int(fx["n_probe"]), np.int64)
    for k in (1, 10, 100):
        v, a = c_oracle.scan_topk(storage, fx["ref_lut"], is_empty, cs, sz, npl, k)
        assert np.array_equal(v, fx[f"orc_vals_k{k}"]) and np.array_equal(a, fx[f"orc_addr_k{k}"])
        vn, an = orc.scan_topk(storage, fx["ref_lut"], is_empty, cs, sz, npl, k)
        assert np.array_equal(vn, v) and np.array_equal(an, a)
    v, a = fx["orc_vals_k100"], fx["orc_addr_k100"]
    tie = (np.diff(v, axis=1) == 0) & (a[:, 1:] >= 0)
    assert tie.sum() > 50
    assert (np.diff(a, axis=1)[tie] > 0).all()           # ties: ascending address
    ids = fx["orc_ids_k100"]
    n = 1500
    # the first 8 queries are stored vectors 0..7: each comes back twice (id q and id n + q), tied
    for q in range(8):
        top2 = set(ids[q, :2].tolist())
        assert top2 == {q, n + q}, (q, top2)
        assert v[q, 0] == v[q, 1]


def test_tombstones_inside_cells(fx_tomb):
    fx = fx_tomb
    storage, is_empty = _sd(fx, "_storage"), _sd(fx, "_is_empty")
    cs = _sd(fx, "_cell_start")[fx["ref_cells"]]
    sz = _sd(fx, "_cell_size")[fx["ref_cells"]]
    npl = np.full(int(fx["nq"]), int(fx["n_probe"]), np.int64)
    dead = fx["dead_address"]
    inside = np.zeros(storage.shape[1], bool)
    for c in range(int(fx["n_cells"])):
        s0 = _sd(fx, "_cell_start")[c]
        inside[s0:s0 + _sd(fx, "_cell_size")[c]] = True
    assert inside[dead].all() and (is_empty[dead] == 1).all()
    for k in (1, 10, 100):
        v, a = c_oracle.scan_topk(storage, fx["ref_lut"], is_empty, cs, sz, npl, k)
        assert np.array_equal(v, fx[f"orc_vals_k{k}"]) and np.array_equal(a, fx[f"orc_addr_k{k}"])
        assert not np.isin(a, dead).any()
        assert np.array_equal(orc.get_id_by_address(_sd(fx, "_address2id"), a), fx[f"orc_ids_k{k}"])
    # without the is_empty test the tombstoned slots would be returned
    v0, a0 = c_oracle.scan_topk(storage, fx["ref_lut"], None, cs, sz, npl, 100)
    assert np.isin(a0, dead).any()


def test_code_layout_round_trip_matches_reference(fx_layout):
    fx = fx_layout
    for case in range(3):
        m = int(fx[f"l{case}_m"])
        ref_storage = fx[f"l{case}_ref_storage"]
        storage = np.zeros_like(ref_storage)
        orc.codes_to_storage(fx[f"l{case}_codes"], fx[f"l{case}_adr"], storage)
        assert storage.shape[0] == m // 4 and np.array_equal(storage, ref_storage)
        got = orc.storage_to_codes(ref_storage, fx[f"l{case}_probe"])
        assert np.array_equal(got, fx[f"l{case}_ref_gather"])


def test_oracle_pinned_against_live_reference():
    """Build container only: run every oracle function that has a runnable counterpart against the
    imported reference itself (oracle/pin_against_reference.py); skipped where /root/reference does
    not exist (the GPU box) -- the committed golden vectors carry the same pins there."""
    from oracle import _refimport
    if not _refimport.available():
        pytest.skip("reference tree not present")
    import warnings
    from oracle import pin_against_reference
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        done = pin_against_reference.run(verbose=False)
    assert len(done) == 7


def _fit_case(fx, case):
    """(init, n_redo, max_iter, tol, seed) of each reference-run fit case in fx_kmeans_fit"""
    return {
        "1": (fx["init"], 1, 1, 0.0, None),
        "3": (fx["init"], 1, 3, 0.0, None),
        "tol": (fx["init"], 1, 12, float(fx["tol_exit"]), None),
        "redo": (fx["init"], 2, 3, 0.0, int(fx["redo_seed"])),
        "redo_b": (fx["bad_init"], 2, 3, 0.0, int(fx["redo_b_seed"])),
    }[case]


@pytest.mark.parametrize("case", ["1", "3", "tol", "redo", "redo_b"])
@pytest.mark.parametrize("numerics", ["direct", "expanded"])
def test_kmeans_fit_driver_matches_reference(fx_kmeans_fit, case, numerics):
    """oracle.kmeans_fit_redo vs the reference's own MultiKMeans.fit run on CPU (fixture made by
    tests/golden/make_golden.py::fx_kmeans_fit): steps, tolerance exit, best-of-n_redo."""
    fx = fx_kmeans_fit
    init, n_redo, max_iter, tol, seed = _fit_case(fx, case)
    if seed is not None:
        np.random.seed(seed)
    cen, labels, inertias, steps = orc.kmeans_fit_redo(
        fx["data"], init.copy(), n_redo, max_iter, tol, init.shape[2],
        assign=lambda a, b: c_oracle.max_sim(a, b, "euclidean", numerics))
    assert (labels == fx[f"ref_labels_{case}"]).mean() >= 0.999
    np.testing.assert_allclose(cen, fx[f"ref_centroids_{case}"], rtol=1e-4, atol=2e-2)
    assert np.mean(np.abs(cen - fx[f"ref_centroids_{case}"]) > 1e-3) < 0.01
    if case == "tol":
        assert steps == [int(fx["tol_exit_steps"])]
    if case in ("redo", "redo_b"):
        ref_in = fx["redo_inertia" if case == "redo" else "redo_b_inertia"]
        np.testing.assert_allclose(inertias, ref_in, rtol=1e-4)
        assert int(np.argmin(inertias)) == int(np.argmin(ref_in)) == (0 if case == "redo" else 1)


def test_cosine_index_against_reference(fx_cosine):
    """distance="cosine" (reference-trained on CPU, make_golden.py::fx_cosine): the oracle's LUT for
    the normalised queries equals the reference's precompute_adc (plain dot products), the scan
    value identity sum_j LUT[j, code_j] == q . decode(code) holds for every stored slot with the
    right-hand side from the reference's decode, and the oracle's top-k are the k largest of those
    among the probed slots."""
    fx = fx_cosine
    pq = _sd(fx, "pq_codec.kmeans.centroids")
    xq = fx["queries_normalized"]
    for lut in (orc.adc_lut(xq, pq, "cosine"), c_oracle.adc_lut(xq, pq, "cosine")):
        np.testing.assert_allclose(lut, fx["ref_lut"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(orc.neg_sq_l2(xq, _sd(fx, "vq_codec.kmeans.centroids")), fx["ref_sims"],
                               rtol=1e-4, atol=1e-5)
    storage, is_empty = _sd(fx, "_storage"), _sd(fx, "_is_empty")
    slots = np.arange(storage.shape[1])
    nq, n_probe = int(fx["nq"]), int(fx["n_probe"])
    for q in range(nq):
        v = orc.scan_values(storage, fx["ref_lut"][:, q, :], slots)
        np.testing.assert_allclose(v, fx["ref_dot_decode"][q], rtol=1e-4, atol=1e-5)
    cs = _sd(fx, "_cell_start")[fx["ref_cells"]]
    sz = _sd(fx, "_cell_size")[fx["ref_cells"]]
    npl = np.full(nq, n_probe, np.int64)
    for k in (1, 10):
        v, a = c_oracle.scan_topk(storage, fx["ref_lut"], is_empty, cs, sz, npl, k)
        assert np.array_equal(v, fx[f"orc_vals_k{k}"]) and np.array_equal(a, fx[f"orc_addr_k{k}"])
        for q in range(nq):
            sl = orc.probed_slots(cs[q], sz[q], npl[q])
            sl = sl[is_empty[sl] == 0]
            exact = np.sort(fx["ref_dot_decode"][q][sl])[::-1][:k]
            np.testing.assert_allclose(v[q][:exact.size], exact, rtol=1e-4, atol=1e-5)


def test_label_disagreement_between_direct_and_expanded_numerics_is_measured():
    """the reference CUDA kernel accumulates fmaf(-(a-b), (a-b), acc)
    (max_sim.cu:78-98, oracle numerics="direct"); the MFMA kernels -- and therefore every cell /
    code that add() stores -- use 2ab - a^2 - b^2 on ascending-k chains (numerics="expanded").  The
    two disagree only where two centroids are within a few ulps of the same distance; this test puts
    a NUMBER on it (INTEGRATION.md 4 quotes the 200 000-point run) and bounds it."""
    from oracle import c_oracle
    rng = np.random.default_rng(0)
    d, n, k = 128, 40000, 512
    cen = np.abs(rng.standard_normal((d, 64))) * 40
    sift = np.clip(np.round(np.abs(cen[:, rng.integers(0, 64, n)] + rng.standard_normal((d, n)) * 25)), 0, 218)
    gist = np.clip(rng.random((d, n)) * 0.6 + rng.standard_normal((d, n)) * 0.1, 0, 1)
    gauss = rng.standard_normal((d, n)) * 3
    rates = {}
    for name, A in (("sift", sift), ("gist", gist), ("gauss", gauss)):
        A = A.astype(np.float32)
        B = A[:, rng.permutation(n)[:k]] + (rng.random((d, k)) * 0.5 if name == "sift"
                                            else rng.standard_normal((d, k)) * 0.01).astype(np.float32)
        v0, i0 = c_oracle.max_sim(A[None], B[None], "euclidean", "direct")
        v1, i1 = c_oracle.max_sim(A[None], B[None], "euclidean", "expanded")
        rates[name] = float((i0 != i1).mean())
        # the maxima agree to fp32 accuracy of the scale |a|^2 + |b|^2 ...
        assert np.abs(v0 - v1).max() <= 2e-5 * np.abs(v0).max()
        # ... and where the labels differ the two candidates are a near-tie under BOTH numerics
        for p in np.nonzero(i0[0] != i1[0])[0]:
            da = -((A[:, p].astype(np.float64) - B[:, i0[0, p]]) ** 2).sum()
            db = -((A[:, p].astype(np.float64) - B[:, i1[0, p]]) ** 2).sum()
            assert abs(da - db) <= 1e-5 * abs(da)
    print("label disagreement direct vs expanded:", rates)
    assert max(rates.values()) <= 1e-3
