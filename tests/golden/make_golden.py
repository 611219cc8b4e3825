"""Generate the golden fixtures in tests/golden/*.npz.

Runs ONLY in the build container: it imports the reference (DeMoriarty/TorchPQ
at /root/reference) through oracle/_refimport.py (stub cupy, CPU code paths)
and records inputs plus the REFERENCE'S OWN outputs for every piece of the
hot path that the reference can execute on CPU.  The fixtures are data (arrays);
the reference's Python never travels.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What is reference-generated (key prefix ``ref_``):
  ref_sims         metric.negative_squared_l2_distance              (metric.py:31-98)
  ref_lut          PQCodec.precompute_adc                           (codec/PQCodec.py:62-75)
  ref_codes/cells  IVFPQIndex.add -> PQCodec.encode / VQCodec.encode (CPU fallbacks;
                   non-negative data, last column dropped: MultiKMeans.py:352-353 bug)
  ref state_dict   buffers after reference train + add              (CellContainer.py:313-367)
  ref_decode       PQCodec.decode                                   (codec/PQCodec.py:113-130)
  ref_adc_exact    -(|q - decode(code)|^2) for every stored slot via reference decode
                   + reference metric: pins the scan's value identity
  ref_nprobe_list  the smart-probing expression of IVFPQIndex.py:499-512 evaluated with torch
  fx_container     CellContainer.add sequences incl. expand()      (CellContainer.py:249-367)
  fx_kmeans        MultiKMeans.get_labels / compute_centroids CPU   (MultiKMeans.py:334-380)
  fx_kmeans_fit    MultiKMeans.fit driver: 1/3 steps, tol exit, n_redo=2 (MultiKMeans.py:415-453)
What is oracle-generated (key prefix ``orc_``): scan top-k (values, addresses, ids) --
the reference has no CPU scan; those are regression vectors for the restatement,
cross-checked in tests against ref_adc_exact.
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle._refimport import import_reference  # noqa: E402
from oracle import ivfpq_oracle as orc  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def sift_like(rng, d, n, n_clusters=40):
    """Non-negative, integer-valued, clustered fp32 data (SIFT-like)."""
    centers = np.abs(rng.standard_normal((d, n_clusters))) * 40
    assign = rng.integers(0, n_clusters, n)
    x = centers[:, assign] + rng.standard_normal((d, n)) * 12
    return np.clip(np.round(np.abs(x)), 0, 218).astype(np.float32)


def build_reference_index(torch, tq, d, m, n_cells, n, initial_size, seed, distance="euclidean"):
    rng = np.random.default_rng(seed)
    base = sift_like(rng, d, n)
    np.random.seed(seed)  # KMeans.initialize_centroids uses np.random.choice (KMeans.py:271-277)
    torch.manual_seed(seed)
    idx = tq.index.IVFPQIndex(d_vector=d, n_subvectors=m, n_cells=n_cells,
                              initial_size=initial_size, distance=distance, device="cpu")
    xb = torch.from_numpy(base.copy())
    idx.train(xb.clone())  # clone: CPU fallback mutates its input (KMeans.py:347)
    return idx, base, rng


def fx_index(torch, tq, name, d, m, n_cells, n, nq, initial_size, seed, n_probe, ks, slim=False):
    """slim=True drops the bulky arrays (base vectors, the all-slots distance matrix) so that a
    larger shape still fits a ~1 MB fixture; the small fixtures pin those."""
    idx, base, rng = build_reference_index(torch, tq, d, m, n_cells, n, initial_size, seed)
    xb = torch.from_numpy(base.copy())
    # reference add; the CPU encode never labels the LAST point (MultiKMeans.py:352-353):
    # append one sacrificial vector and remember that its code column is not pinned.
    ids = idx.add(xb.clone())
    sd = {k: v.numpy().copy() for k, v in idx.state_dict().items() if v is not None}
    queries = sift_like(rng, d, nq)
    xq = torch.from_numpy(queries.copy())

    vq_cb = idx.vq_codec.codebook
    pq_cb = idx.pq_codec.codebook
    ref_sims = tq.metric.negative_squared_l2_distance(xq.clone(), vq_cb.clone()).numpy()
    ref_lut = idx.pq_codec.precompute_adc(xq.clone()).numpy()

    # smart probing expression (IVFPQIndex.py:499-512) on the reference sims
    sims_t = torch.from_numpy(ref_sims)
    topk_sims, cells = sims_t.topk(n_probe, dim=1)
    p = -topk_sims.abs().sqrt()
    p = torch.softmax(p / 30.0, dim=-1)
    max_n_probe = torch.tensor(n_probe)
    ne = -torch.sum(p * torch.log2(p) / torch.log2(max_n_probe), dim=-1)
    ref_npl = torch.ceil(ne * max_n_probe).long().numpy()

    # decode identity for every stored slot
    cap = idx._storage.shape[1]
    all_adr = torch.arange(cap)
    codes_all = idx.get_data_by_address(all_adr.clone())  # [m, cap]
    ref_decode = idx.pq_codec.decode(codes_all).numpy()  # [d, cap]
    ref_adc_exact = tq.metric.negative_squared_l2_distance(
        xq.clone(), torch.from_numpy(ref_decode.copy())).numpy()  # [nq, cap]

    out = dict(
        d=d, m=m, n_cells=n_cells, n=n, nq=nq, n_probe=n_probe, ks=np.array(ks),
        base=base, queries=queries, add_ids=ids.numpy(),
        ref_sims=ref_sims, ref_lut=ref_lut, ref_topk_sims=topk_sims.numpy(),
        ref_cells=cells.numpy(), ref_nprobe_list=ref_npl,
        ref_decode_codes=codes_all.numpy()[:, :256].copy(),
        ref_decode=ref_decode[:, :256].copy(),
        ref_adc_exact=ref_adc_exact,
    )
    if slim:
        for key in ("base", "ref_adc_exact", "add_ids"):
            out.pop(key)
    for k, v in sd.items():
        out["sd." + k] = v

    # oracle-generated regression vectors for the scan
    storage = sd["_storage"]
    is_empty = sd["_is_empty"]
    cs = sd["_cell_start"][out["ref_cells"]]
    sz = sd["_cell_size"][out["ref_cells"]]
    lut = ref_lut
    for smart in (0, 1):
        npl = ref_npl if smart else np.full(nq, n_probe, np.int64)
        for k in ks:
            v, a = orc.scan_topk(storage, lut, is_empty, cs, sz, npl, int(k))
            out[f"orc_vals_s{smart}_k{k}"] = v
            out[f"orc_addr_s{smart}_k{k}"] = a
            out[f"orc_ids_s{smart}_k{k}"] = orc.get_id_by_address(sd["_address2id"], a)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(name, "n_items", int(sd["_cell_size"].sum()), "cap", cap)


def fx_container(torch, tq):
    """CellContainer.add sequences (with expand) run by the reference on CPU."""
    rng = np.random.default_rng(7)
    out = {}
    for case, (mode, step) in enumerate([("double", 8), ("step", 8)]):
        c = tq.container.CellContainer(code_size=8, n_cells=5, dtype="uint8", device="cpu",
                                       initial_size=4, expand_step_size=step, expand_mode=mode,
                                       use_inverse_id_mapping=True, contiguous_size=4)
        batches = []
        for b, nb in enumerate([6, 17, 3, 40]):
            codes = rng.integers(0, 256, (8, nb), dtype=np.uint8)
            cells = rng.integers(0, 5, nb).astype(np.int64)
            if b == 2:
                ids = (1000 + np.arange(nb)).astype(np.int64)
                r_ids, r_adr = c.add(torch.from_numpy(codes), torch.from_numpy(cells),
                                     ids=torch.from_numpy(ids), return_address=True)
            else:
                ids = None
                r_ids, r_adr = c.add(torch.from_numpy(codes), torch.from_numpy(cells),
                                     return_address=True)
            out[f"c{case}_b{b}_codes"] = codes
            out[f"c{case}_b{b}_cells"] = cells
            out[f"c{case}_b{b}_ids_in"] = ids if ids is not None else np.zeros(0, np.int64)
            out[f"c{case}_b{b}_ref_ids"] = r_ids.numpy()
            out[f"c{case}_b{b}_ref_adr"] = r_adr.numpy()
            out[f"c{case}_b{b}_ref_ioa"] = c.get_ioa(torch.from_numpy(cells)).numpy()
            for k in ["_storage", "_cell_start", "_cell_size", "_cell_capacity", "_is_empty",
                      "_address2id"]:
                out[f"c{case}_b{b}_sd{k}"] = getattr(c, k).numpy().copy()
            out[f"c{case}_b{b}_max_id"] = np.int64(c.max_id)
        probe = torch.arange(-2, c.capacity + 2)
        out[f"c{case}_probe_adr"] = probe.numpy()
        out[f"c{case}_ref_cell_of_adr"] = c.get_cell_by_address(probe.clone()).numpy()
        out[f"c{case}_ref_id_of_adr"] = c.get_id_by_address(probe.clone()).numpy()
        out[f"c{case}_ref_data_of_adr"] = c.get_data_by_address(probe.clone()).numpy()
        out[f"c{case}_mode"] = np.array([mode == "double", step])
    np.savez_compressed(os.path.join(OUT, "fx_container.npz"), **out)
    print("fx_container ok")


def fx_kmeans(torch, tq):
    """MultiKMeans CPU assign/update on non-negative data with fixed centroids."""
    rng = np.random.default_rng(11)
    l, d, n, k = 4, 8, 4096, 32
    data = np.stack([sift_like(rng, d, n, 12) for _ in range(l)])  # [l, d, n], >= 0
    init = data[:, :, rng.choice(n, k, replace=False)].copy()
    mk = tq.clustering.MultiKMeans(n_clusters=k, distance="euclidean", max_iter=3)
    d_t = torch.from_numpy(data.copy())
    c_t = torch.from_numpy(init.copy())
    # CPU get_labels mutates (abs) its inputs -> clones of non-negative arrays; last point never
    # labelled (MultiKMeans.py:352-353) -> stored without the last column.
    ms, lab = mk.get_labels(d_t.clone(), c_t.clone())
    ref_labels = lab.numpy()[:, :-1].copy()
    ref_maxsims = ms.numpy()[:, :-1].copy()
    # exact-label update through the reference loop (MultiKMeans.py:367-380)
    full_labels = orc.max_sim(data, init, "euclidean", "expanded")[1]
    ref_centroids = mk.compute_centroids(d_t.clone(), torch.from_numpy(full_labels)).numpy()
    sims = mk.euc_sim(d_t.clone(), c_t.clone()).numpy()  # [l, n, k]
    np.savez_compressed(os.path.join(OUT, "fx_kmeans.npz"), data=data, init=init,
                        ref_labels=ref_labels, ref_maxsims=ref_maxsims,
                        labels_for_update=full_labels, ref_centroids=ref_centroids,
                        ref_sims_top2gap=np.sort(sims, axis=-1)[..., -1] - np.sort(sims, axis=-1)[..., -2])
    print("fx_kmeans ok")


def fx_kmeans_fit(torch, tq):
    """The reference's own Lloyd DRIVER (MultiKMeans.fit, MultiKMeans.py:415-453) run on CPU from
    fixed initial centroids: 1 step, 3 steps, a tolerance early exit, and n_redo=2 (best-inertia
    selection, second redo initialised by np.random.choice as in :277-283).

    The CPU branch of get_labels never labels the last point and abs()-es its inputs
    (MultiKMeans.py:352-358); `fit` is therefore run with get_labels bound to the reference's
    own in-memory branch (:311-313, disabled upstream by `if False`): sim(inplace=False) + max.
    Everything else -- compute_centroids_loop, calculate_error, calculate_inertia, the tol test,
    the redo bookkeeping, initialize_centroids -- is the unmodified reference code."""
    import types
    rng = np.random.default_rng(21)
    l, d, n, k = 3, 8, 3000, 24
    data = np.stack([sift_like(rng, d, n, 10) for _ in range(l)])  # [l, d, n] >= 0
    init = data[:, :, rng.choice(n, k, replace=False)].copy()
    out = {"data": data, "init": init}

    def in_memory_get_labels(self, data, centroids):
        sims = self.sim(data, centroids, inplace=False)
        return sims.max(dim=-1)

    def run(max_iter, tol, n_redo=1, seed=None, verbose_log=None, init=init):
        mk = tq.clustering.MultiKMeans(n_clusters=k, distance="euclidean", max_iter=max_iter,
                                       tol=tol, n_redo=n_redo)
        mk.get_labels = types.MethodType(in_memory_get_labels, mk)
        if verbose_log is not None:  # record (error, inertia) per iteration through the
            mk.verbose = 3           # reference's own progress messages (:433)
            mk.print_message = lambda msg, level=1: verbose_log.append(msg)
        if seed is not None:
            np.random.seed(seed)
        labels = mk.fit(torch.from_numpy(data.copy()), torch.from_numpy(init.copy()))
        return mk.centroids.numpy().copy(), labels.numpy().copy()

    for steps in (1, 3):
        c, lab = run(steps, 0.0)
        out[f"ref_centroids_{steps}"] = c
        out[f"ref_labels_{steps}"] = lab
    # tolerance exit: errors of a 12-step run, then tol halfway between the errors of steps 3 and 4
    log = []
    run(12, 0.0, verbose_log=log)
    errs = [float(m.split("error=")[1].split(",")[0]) for m in log if "iteration" in m]
    assert len(errs) == 12 and errs[2] > errs[3] > 0, errs
    tol = 0.5 * (errs[2] + errs[3])
    c, lab = run(12, tol)
    c4, lab4 = run(4, 0.0)
    assert np.array_equal(c, c4) and np.array_equal(lab, lab4), "tol exit must stop after step 4"
    out["errors_12"] = np.array(errs, dtype=np.float64)
    out["tol_exit"] = np.float64(tol)
    out["tol_exit_steps"] = np.int64(4)
    out["ref_centroids_tol"] = c
    out["ref_labels_tol"] = lab
    # n_redo = 2: redo 0 from `init`, redo 1 from np.random.choice under seed 77
    log = []
    c, lab = run(3, 0.0, n_redo=2, seed=77, verbose_log=log)
    inert = [float(m.split("inertia: ")[1].split("time")[0]) for m in log if "redo finished" in m]
    assert len(inert) == 2
    out["redo_seed"] = np.int64(77)
    out["redo_inertia"] = np.array(inert, dtype=np.float64)
    out["ref_centroids_redo"] = c
    out["ref_labels_redo"] = lab
    # ... and from a poor start (all initial centroids within +-1 of one point) so that redo 1 wins
    bad = (data[:, :, :1] + rng.integers(0, 2, (l, d, k))).astype(np.float32)
    log = []
    c, lab = run(3, 0.0, n_redo=2, seed=78, verbose_log=log, init=bad)
    inert = [float(m.split("inertia: ")[1].split("time")[0]) for m in log if "redo finished" in m]
    assert len(inert) == 2 and inert[1] < inert[0], inert
    out["bad_init"] = bad
    out["redo_b_seed"] = np.int64(78)
    out["redo_b_inertia"] = np.array(inert, dtype=np.float64)
    out["ref_centroids_redo_b"] = c
    out["ref_labels_redo_b"] = lab
    np.savez_compressed(os.path.join(OUT, "fx_kmeans_fit.npz"), **out)
    print("fx_kmeans_fit ok: errors", [round(e, 3) for e in errs[:5]], "redo inertia", inert)


def fx_cosine(torch, tq):
    """distance="cosine": reference train / add on CPU (IVFPQIndex.py:234-260,316-364: inputs
    normalised, coarse quantiser euclidean on the normalised vectors, PQ k-means and the ADC table
    by dot product -- codec/PQCodec.py:62-75 with cos_sim(normalize=False)), the reference's coarse
    sims and LUT for normalised queries; scan results from the oracle (no CPU scan upstream)."""
    d, m, n_cells, n, nq, n_probe = 32, 8, 16, 2500, 12, 4
    rng = np.random.default_rng(8)
    base = sift_like(rng, d, n)
    np.random.seed(8)
    torch.manual_seed(8)
    idx = tq.index.IVFPQIndex(d_vector=d, n_subvectors=m, n_cells=n_cells, initial_size=256,
                              distance="cosine", device="cpu")
    xb = torch.from_numpy(base.copy())
    idx.train(xb.clone())
    ids = idx.add(xb.clone())
    sd = {k: v.numpy().copy() for k, v in idx.state_dict().items() if v is not None}
    queries = sift_like(rng, d, nq)
    xq = tq.util.normalize(torch.from_numpy(queries.copy()), dim=0)  # search() normalises (:480-481)
    ref_sims = tq.metric.negative_squared_l2_distance(xq.clone(), idx.vq_codec.codebook.clone()).numpy()
    ref_lut = idx.pq_codec.precompute_adc(xq.clone()).numpy()
    topk_sims, cells = torch.from_numpy(ref_sims).topk(n_probe, dim=1)
    # every stored code's value through the reference's decode: sum_j LUT == q . decode(code)
    cap = idx._storage.shape[1]
    codes_all = idx.get_data_by_address(torch.arange(cap))
    ref_decode = idx.pq_codec.decode(codes_all).numpy()
    ref_dot = (xq.numpy().T.astype(np.float64) @ ref_decode.astype(np.float64)).astype(np.float32)
    out = dict(d=d, m=m, n_cells=n_cells, n=n, nq=nq, n_probe=n_probe, ks=np.array([1, 10]),
               base=base, queries=queries, queries_normalized=xq.numpy(), add_ids=ids.numpy(),
               ref_sims=ref_sims, ref_lut=ref_lut, ref_cells=cells.numpy(), ref_dot_decode=ref_dot)
    for k, v in sd.items():
        out["sd." + k] = v
    cs = sd["_cell_start"][out["ref_cells"]]
    sz = sd["_cell_size"][out["ref_cells"]]
    npl = np.full(nq, n_probe, np.int64)
    for k in (1, 10):
        v, a = orc.scan_topk(sd["_storage"], ref_lut, sd["_is_empty"], cs, sz, npl, k)
        out[f"orc_vals_k{k}"] = v
        out[f"orc_addr_k{k}"] = a
        out[f"orc_ids_k{k}"] = orc.get_id_by_address(sd["_address2id"], a)
    np.savez_compressed(os.path.join(OUT, "fx_cosine.npz"), **out)
    print("fx_cosine ok: n_items", int(sd["_cell_size"].sum()), "lut range", float(ref_lut.min()), float(ref_lut.max()))


def fx_residual(torch, tq):
    """pq_use_residual=True: reference train/add on CPU, reference part1/part2/full tables
    (IVFPQIndex.py:160-170, 366-405); scan results from the oracle (no CPU scan in the reference)."""
    d, m, n_cells, n, nq, n_probe = 32, 8, 16, 3000, 12, 4
    rng = np.random.default_rng(5)
    base = sift_like(rng, d, n)
    np.random.seed(5)
    idx = tq.index.IVFPQIndex(d_vector=d, n_subvectors=m, n_cells=n_cells, initial_size=256,
                              device="cpu", pq_use_residual=True)
    idx.train(torch.from_numpy(base.copy()))
    idx.add(torch.from_numpy(base.copy()))
    sd = {k: v.numpy().copy() for k, v in idx.state_dict().items() if v is not None}
    queries = sift_like(rng, d, nq)
    xq = torch.from_numpy(queries.copy())
    sims = tq.metric.negative_squared_l2_distance(xq.clone(), idx.vq_codec.codebook.clone())
    topk_sims, cells = sims.topk(n_probe, dim=1)
    part1, part2 = idx.precomputed_adc_residual_precomputed(xq.clone())
    full = idx.precomputed_adc_residual(xq.clone(), cells)
    part1, part2, full = (np.ascontiguousarray(t.numpy()) for t in (part1, part2, full))
    out = dict(d=d, m=m, n_cells=n_cells, n=n, nq=nq, n_probe=n_probe, base=base, queries=queries,
               ref_topk_sims=topk_sims.numpy(), ref_cells=cells.numpy(), ref_part1=part1,
               ref_part2=part2, ref_full=full)
    for k, v in sd.items():
        out["sd." + k] = v
    # value identity through reference functions: base + sum_j LUT == -|q - (centroid + decode)|^2
    cap = idx._storage.shape[1]
    codes_all = idx.get_data_by_address(torch.arange(cap))
    cell_of_slot = idx.get_cell_by_address(torch.arange(cap)).clamp(min=0)
    recon = idx.decode((codes_all, cell_of_slot)).numpy()
    out["ref_adc_exact"] = tq.metric.negative_squared_l2_distance(
        xq.clone(), torch.from_numpy(recon.copy())).numpy()
    cs, sz = sd["_cell_start"][out["ref_cells"]], sd["_cell_size"][out["ref_cells"]]
    npl = np.full(nq, n_probe, np.int64)
    for k in (1, 10, 100):
        v, a = orc.scan_topk_residual(sd["_storage"], part1, part2, out["ref_cells"],
                                      out["ref_topk_sims"], sd["_is_empty"], cs, sz, npl, k)
        out[f"orc_vals_k{k}"], out[f"orc_addr_k{k}"] = v, a
        v2, a2 = orc.scan_topk_residual(sd["_storage"], None, None, None, out["ref_topk_sims"],
                                        sd["_is_empty"], cs, sz, npl, k, full=full)
        out[f"orc_full_vals_k{k}"], out[f"orc_full_addr_k{k}"] = v2, a2
    np.savez_compressed(os.path.join(OUT, "fx_residual.npz"), **out)
    print("fx_residual ok", int(sd["_cell_size"].sum()))


def fx_ties_and_tomb(torch, tq):
    """fx_ties: the reference adds 400 vectors a second time (new ids): identical codes in the same
    cell -> exact value ties; pins the tie policy (value desc, address asc).
    fx_tomb: the same index with tombstones INSIDE the cells' occupied ranges.  The reference's
    remove() never gets past its inverted guard (CellContainer.py:381-383), so the two buffer
    writes it would make (`_is_empty[address] = 1; _address2id[address] = -1`, :387-388) are
    applied to the reference object's buffers directly; `_cell_size` is left as is so that the
    scan's per-slot `is_empty` test (ivfpq_topk.cu:883-884) is what removes them."""
    d, m, n_cells, n, nq, n_probe = 32, 8, 16, 1500, 16, 6
    idx, base, rng = build_reference_index(torch, tq, d, m, n_cells, n, 256, seed=21)
    idx.add(torch.from_numpy(base.copy()))
    idx.add(torch.from_numpy(base[:, :400].copy()))          # duplicates, ids n .. n+399
    queries = np.concatenate([base[:, :8], sift_like(rng, d, nq - 8)], axis=1)  # 8 queries ARE stored vectors
    xq = torch.from_numpy(queries.copy())
    sims = tq.metric.negative_squared_l2_distance(xq.clone(), idx.vq_codec.codebook.clone())
    topk_sims, cells = sims.topk(n_probe, dim=1)
    lut = idx.pq_codec.precompute_adc(xq.clone()).numpy()
    npl = np.full(nq, n_probe, np.int64)

    def dump(name, extra):
        sd = {k: v.numpy().copy() for k, v in idx.state_dict().items() if v is not None}
        out = dict(d=d, m=m, n_cells=n_cells, nq=nq, n_probe=n_probe, ks=np.array([1, 10, 100]),
                   queries=queries, ref_lut=lut, ref_cells=cells.numpy(),
                   ref_topk_sims=topk_sims.numpy(), **extra)
        for k, v in sd.items():
            out["sd." + k] = v
        cs, sz = sd["_cell_start"][out["ref_cells"]], sd["_cell_size"][out["ref_cells"]]
        for k in (1, 10, 100):
            v, a = orc.scan_topk(sd["_storage"], lut, sd["_is_empty"], cs, sz, npl, k)
            out[f"orc_vals_k{k}"], out[f"orc_addr_k{k}"] = v, a
            out[f"orc_ids_k{k}"] = orc.get_id_by_address(sd["_address2id"], a)
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
        return out

    t = dump("fx_ties", {})
    ties = int((np.diff(t["orc_vals_k100"], axis=1) == 0).sum())
    assert ties > 50, ties
    print("fx_ties ok, exact ties inside the top-100 lists:", ties)

    occupied = np.nonzero(idx._is_empty.numpy() == 0)[0]
    dead = torch.from_numpy(np.sort(rng.choice(occupied, 300, replace=False)))
    dead_ids = idx._address2id[dead].clone()
    idx._is_empty[dead] = 1          # CellContainer.py:387
    idx._address2id[dead] = -1       # CellContainer.py:388
    t = dump("fx_tomb", dict(dead_address=dead.numpy(), dead_ids=dead_ids.numpy()))
    assert not np.isin(t["orc_addr_k100"], dead.numpy()).any()
    print("fx_tomb ok")


def fx_layout(torch, tq):
    """[m, n] codes <-> _storage [m/4, cap, 4] through the reference's own
    set_data_by_address / get_data_by_address (CellContainer.py:151-239)."""
    rng = np.random.default_rng(31)
    out = {}
    for case, (m, cap_cells) in enumerate([(8, 5), (16, 3), (120, 2)]):
        c = tq.container.CellContainer(code_size=m, n_cells=cap_cells, dtype="uint8", device="cpu",
                                       initial_size=16, expand_step_size=8, expand_mode="double",
                                       use_inverse_id_mapping=True, contiguous_size=4)
        cap = c.capacity
        nw = min(40, cap - 3)
        adr = rng.choice(cap, nw, replace=False).astype(np.int64)
        codes = rng.integers(0, 256, (m, nw), dtype=np.uint8)
        c.set_data_by_address(torch.from_numpy(codes), torch.from_numpy(adr))
        probe = np.concatenate([adr[::-1], np.array([-1, cap, cap + 7]), rng.integers(0, cap, 9)])
        out[f"l{case}_m"] = np.int64(m)
        out[f"l{case}_codes"] = codes
        out[f"l{case}_adr"] = adr
        out[f"l{case}_ref_storage"] = c._storage.numpy().copy()
        out[f"l{case}_probe"] = probe.astype(np.int64)
        out[f"l{case}_ref_gather"] = c.get_data_by_address(torch.from_numpy(probe.astype(np.int64))).numpy()
    np.savez_compressed(os.path.join(OUT, "fx_layout.npz"), **out)
    print("fx_layout ok")


def main():
    tq = import_reference()
    import torch
    torch.set_num_threads(4)
    jobs = {
        "fx_tiny": lambda: fx_index(torch, tq, "fx_tiny", d=32, m=8, n_cells=16, n=2000, nq=16,
                                    initial_size=256, seed=1, n_probe=4, ks=[1, 10, 100]),
        "fx_m16": lambda: fx_index(torch, tq, "fx_m16", d=64, m=16, n_cells=32, n=6000, nq=24,
                                   initial_size=128, seed=2, n_probe=8, ks=[10]),
        # BASELINE.json configs[0] shape (d=128, m=16, n_cells=256, nprobe=8, k=10), 20 000 of its
        # 100 000 vectors so that the fixture stays ~1 MB
        "fx_c1": lambda: fx_index(torch, tq, "fx_c1", d=128, m=16, n_cells=256, n=20000, nq=64,
                                  initial_size=128, seed=3, n_probe=8, ks=[10], slim=True),
        "fx_container": lambda: fx_container(torch, tq),
        "fx_kmeans": lambda: fx_kmeans(torch, tq),
        "fx_kmeans_fit": lambda: fx_kmeans_fit(torch, tq),
        "fx_residual": lambda: fx_residual(torch, tq),
        "fx_cosine": lambda: fx_cosine(torch, tq),
        "fx_ties_tomb": lambda: fx_ties_and_tomb(torch, tq),
        "fx_layout": lambda: fx_layout(torch, tq),
    }
    names = sys.argv[1:] or list(jobs)
    for n in names:
        jobs[n]()


if __name__ == "__main__":
    main()
