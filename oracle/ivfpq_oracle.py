"""CPU oracle for the IVFPQ train / add / search hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``torchpq_amd/`` may import this
module: it exists so that ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` can check (and time) the HIP path against
a plain restatement of the reference's algorithm.

Every function restates one piece of DeMoriarty/TorchPQ (reference paths are
relative to the reference repository root) in numpy; byte / integer work is
bit-exact, floating point follows the reference's formula and summation order
where the reference fixes one.

Pinning status (see oracle/pin_against_reference.py, tests/golden/):
  * coarse sims, ADC LUT, smart probing, encode labels, container placement,
    decode, id mapping: PINNED against the imported reference Python (stub
    cupy) in the build container; golden vectors committed.
  * list scan (``scan_topk``): the reference has no CPU implementation and no
    test/golden vector for it ("parity unpinned" by the reference's own
    tests).  It is pinned indirectly through reference functions executed
    here: sum_j LUT[j, q, code_j] == -(|q - decode(code)|^2) with LUT from the
    reference's ``precompute_adc`` and ``decode`` from the reference, and by
    the kernel text ivfpq_topk.cu:662-679 (ascending-j fp32 accumulation).
"""
from __future__ import annotations

import numpy as np

F32 = np.float32
NEG_INF = F32(-np.inf)


# --------------------------------------------------------------------------
# coarse probe
# --------------------------------------------------------------------------
def neg_sq_l2(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """``2 a^T b - |a|^2 - |b|^2`` in fp32.

    a: [d, m], b: [d, n] -> [m, n].
    Restates torchpq/metric.py:75-98 (non tensor-core branch): GEMM, ``*2``,
    ``- sum(a**2)``, ``- sum(b**2)`` in that order.
    """
    a = np.ascontiguousarray(a, dtype=F32)
    b = np.ascontiguousarray(b, dtype=F32)
    y = (a.T @ b).astype(F32)
    y *= F32(2)
    y -= (a * a).sum(axis=0, dtype=F32)[:, None]
    y -= (b * b).sum(axis=0, dtype=F32)[None, :]
    return y


def topk_desc(x: np.ndarray, k: int):
    """Row-wise top-k, sorted descending; ties -> smaller column first.

    Restates the *contract* of torchpq/fn/Topk.py:43-67 (Top32Select /
    TopkSelect / Top1Select: values descending, int64 indices).  The
    reference's bitonic network duplicates one index and loses the other on
    exact ties (top32_select.cu:42-57); the build returns distinct indices,
    lower index first, which is what this oracle defines.
    """
    x = np.asarray(x)
    rows, cols = x.shape
    assert 1 <= k <= cols
    # stable sort on -x keeps the lower index first among equal values
    order = np.argsort(-x, axis=1, kind="stable")[:, :k]
    vals = np.take_along_axis(x, order, axis=1)
    return vals.astype(x.dtype), order.astype(np.int64)


def smart_probing(topk_sims: np.ndarray, n_probe: int, temperature: float = 30.0):
    """n_probe_list from the entropy of the coarse similarities.

    Restates torchpq/index/IVFPQIndex.py:499-512:
      p = softmax(-sqrt(|s|) / T); H = -sum(p log2 p / log2 n_probe);
      n_probe_list = ceil(H * n_probe).long()
    """
    s = np.asarray(topk_sims, dtype=F32)
    p = -np.sqrt(np.abs(s))
    z = p / F32(temperature)
    z = z - z.max(axis=-1, keepdims=True)
    e = np.exp(z, dtype=F32)
    p = e / e.sum(axis=-1, keepdims=True, dtype=F32)
    with np.errstate(divide="ignore", invalid="ignore"):
        h = -(p * np.log2(p) / np.log2(F32(n_probe))).sum(axis=-1, dtype=F32)
    return np.ceil(h * F32(n_probe)).astype(np.int64)


# --------------------------------------------------------------------------
# ADC look-up table
# --------------------------------------------------------------------------
def adc_lut(query: np.ndarray, codebook: np.ndarray, distance: str = "euclidean"):
    """Per-query asymmetric-distance LUT, fp32 ``[m, nq, 256]``.

    query [d, nq], codebook [m, ds, 256].
    Restates torchpq/codec/PQCodec.py:62-75 -> MultiKMeans.sim
    (clustering/MultiKMeans.py:211-223): euclidean -> euc_sim :184-209
    (``2 q.c - |q|^2 - |c|^2``), cosine/inner -> plain dot (:155-181 with
    normalize=False).
    """
    m, ds, k = codebook.shape
    d, nq = query.shape
    assert d == m * ds
    q = np.ascontiguousarray(query, dtype=F32).reshape(m, ds, nq)
    c = np.ascontiguousarray(codebook, dtype=F32)
    y = np.einsum("jdq,jdc->jqc", q, c).astype(F32)
    if distance == "euclidean":
        y *= F32(2)
        y -= (q * q).sum(axis=1, dtype=F32)[:, :, None]
        y -= (c * c).sum(axis=1, dtype=F32)[:, None, :]
    elif distance in ("cosine", "inner"):
        pass
    else:
        raise ValueError(distance)
    return y


# --------------------------------------------------------------------------
# storage layout
# --------------------------------------------------------------------------
def codes_to_storage(codes: np.ndarray, address: np.ndarray, storage: np.ndarray):
    """Scatter codes [m, n] into ``_storage`` [m/4, cap, 4] at ``address``.

    Restates CellContainer.set_data_by_address (container/CellContainer.py:213-239).
    """
    m, n = codes.shape
    g = storage.shape[0]
    cs = storage.shape[2]
    assert g * cs == m
    data = codes.reshape(g, cs, n).transpose(0, 2, 1)  # [g, n, cs]
    mask = (address >= 0) & (address < storage.shape[1])
    storage[:, address[mask]] = data[:, mask]


def storage_to_codes(storage: np.ndarray, address: np.ndarray):
    """Gather codes [m, n] from ``_storage`` (CellContainer.py:151-211);
    invalid addresses give zero columns."""
    g, cap, cs = storage.shape
    mask = (address >= 0) & (address < cap)
    adr = np.where(mask, address, 0)
    data = storage[:, adr]  # [g, n, cs]
    data = data.copy()
    data[:, ~mask] = 0
    return data.transpose(0, 2, 1).reshape(g * cs, -1)


# --------------------------------------------------------------------------
# the list scan (a-1 / a-2)
# --------------------------------------------------------------------------
def scan_values(storage, lut_q, slots):
    """ADC value of every slot in ``slots`` for one query.

    ``v = 0.f; for j ascending: v += LUT[j][code_j]`` in fp32 -- the order of
    consume_data (kernels/cuda/ivfpq_topk.cu:662-679).
    storage [m/4, cap, 4] u8, lut_q [m, 256] f32.
    """
    g, cap, cs = storage.shape
    m = g * cs
    v = np.zeros(slots.shape[0], dtype=F32)
    for j in range(m):
        c = storage[j // cs, slots, j % cs]
        v = (v + lut_q[j, c]).astype(F32)
    return v


def probed_slots(cell_start_q, cell_size_q, n_probe):
    """Concatenated slot addresses of the first ``n_probe`` probed cells.

    Restates the cell walk of ivfpq_topk.cu:856-870 including its guard that
    skips a cell whose start equals the previous cell's start (:864-866).
    """
    out = []
    prev_start = None
    for p in range(int(n_probe)):
        st = int(cell_start_q[p])
        sz = int(cell_size_q[p])
        if prev_start is not None and st == prev_start:
            prev_start = st
            continue
        prev_start = st
        if sz > 0:
            out.append(np.arange(st, st + sz, dtype=np.int64))
    if not out:
        return np.zeros(0, dtype=np.int64)
    return np.concatenate(out)


def scan_topk(storage, lut, is_empty, cell_start, cell_size, n_probe_list, k,
              base_sims=None):
    """IVF list scan + top-k for a batch of queries.

    storage u8 [m/4, cap, 4]; lut f32 [m, nq, 256]; is_empty u8 [cap] or None;
    cell_start / cell_size i64 [nq, max_n_probe]; n_probe_list i64 [nq].
    Returns (values f32 [nq, k] descending, address i64 [nq, k]); unfilled
    positions are (-inf, -1).

    Restates ivfpq_topk.cu:822-971 + IVFPQTopkCuda.topk
    (kernels/IVFPQTopkCuda.py:81-142): candidates are the non-tombstoned
    slots (:883-884) of the probed cells; output sorted by value descending.
    Exact ties are ordered by ascending address (the reference's network is
    order-unstable and may duplicate ids on ties -- SURVEY 7.1).
    """
    nq = cell_start.shape[0]
    vals = np.full((nq, k), NEG_INF, dtype=F32)
    adr = np.full((nq, k), -1, dtype=np.int64)
    for q in range(nq):
        slots = probed_slots(cell_start[q], cell_size[q], n_probe_list[q])
        if is_empty is not None and slots.size:
            slots = slots[is_empty[slots] == 0]
        if slots.size == 0:
            continue
        v = scan_values(storage, lut[:, q, :], slots)
        order = np.lexsort((slots, -v))[:k]
        n = order.size
        vals[q, :n] = v[order]
        adr[q, :n] = slots[order]
    return vals, adr


def scan_topk_residual(storage, part1, part2, cells, base_sims, is_empty, cell_start, cell_size,
                       n_probe_list, k, full=None):
    """Residual-PQ list scan.  value(slot in probe p) = base_sims[q, p], then for j ascending
    ``+= LUT_p[j][code_j]`` with ``LUT_p = part1[q] + part2[cells[q, p]]`` (one fp32 add per entry)
    -- ivfpq_topk_residual_precomputed (kernels/cuda/ivfpq_topk.cu:1039-1208, LUT build
    load_precomputed_v3 :522-560, accumulation start ``newPair.value = cBaseSim`` :1113) -- or,
    when ``full`` [nq, n_probe, m, 256] is given, ``LUT_p = full[q, p]`` (ivfpq_topk_residual
    :973-1037).  part1 [nq, m, 256], part2 [n_cells, m, 256]; logical indexing (the reference hands
    its kernel permuted views)."""
    nq = cell_start.shape[0]
    vals = np.full((nq, k), NEG_INF, dtype=F32)
    adr = np.full((nq, k), -1, dtype=np.int64)
    g, cap, cs4 = storage.shape
    m = g * cs4
    for q in range(nq):
        all_v, all_s = [], []
        prev = None
        for p in range(int(n_probe_list[q])):
            st, sz = int(cell_start[q, p]), int(cell_size[q, p])
            if prev is not None and st == prev:
                prev = st
                continue
            prev = st
            if sz <= 0:
                continue
            slots = np.arange(st, st + sz, dtype=np.int64)
            if is_empty is not None:
                slots = slots[is_empty[slots] == 0]
            if slots.size == 0:
                continue
            if full is not None:
                lut_p = full[q, p].astype(F32)
            else:
                lut_p = (part1[q].astype(F32) + part2[int(cells[q, p])].astype(F32)).astype(F32)
            v = np.full(slots.shape[0], F32(base_sims[q, p]), dtype=F32)
            for j in range(m):
                c = storage[j // cs4, slots, j % cs4]
                v = (v + lut_p[j, c]).astype(F32)
            all_v.append(v)
            all_s.append(slots)
        if not all_v:
            continue
        v = np.concatenate(all_v)
        sl = np.concatenate(all_s)
        order = np.lexsort((sl, -v))[:k]
        vals[q, :order.size] = v[order]
        adr[q, :order.size] = sl[order]
    return vals, adr


def residual_part1(query, pq_codebook):
    """part1[q, j, c] = 2 * q_j . r_jc  (IVFPQIndex.precomputed_adc_residual_precomputed,
    index/IVFPQIndex.py:366-379), dots as ascending-dimension fma chains."""
    m, ds, _ = pq_codebook.shape
    d, nq = query.shape
    qs = np.ascontiguousarray(query, dtype=F32).reshape(m, ds, nq)
    out = np.zeros((nq, m, 256), dtype=F32)
    for e in range(ds):
        out = _fma(qs[:, e, :].T[:, :, None], pq_codebook[None, :, e, :], out)
    return (F32(2) * out).astype(F32)


def residual_part2(vq_codebook, pq_codebook):
    """part2[cell, j, c] = -2 * c_j . r_jc - |r_jc|^2  (IVFPQIndex.precompute_part2,
    index/IVFPQIndex.py:160-170); fp32, BLAS summation order -> compare with tolerance."""
    m, ds, _ = pq_codebook.shape
    n_cells = vq_codebook.shape[1]
    vq = np.ascontiguousarray(vq_codebook, dtype=F32).reshape(m, ds, n_cells)
    prod = np.einsum("jdn,jdc->jnc", vq, pq_codebook).astype(F32)
    nrm = (pq_codebook * pq_codebook).sum(axis=1, dtype=F32)  # [m, 256]
    out = (prod * F32(-2) - nrm[:, None, :]).astype(F32)
    return np.ascontiguousarray(out.transpose(1, 0, 2))


def get_id_by_address(address2id, address):
    """BaseContainer.get_id_by_address (container/BaseContainer.py:58-65)."""
    mask = (address >= 0) & (address < address2id.shape[0])
    ids = np.full(address.shape, -1, dtype=np.int64)
    ids[mask] = address2id[address[mask]]
    return ids


# --------------------------------------------------------------------------
# k-means assign / update (a-8, a-9)
# --------------------------------------------------------------------------
def max_sim(A, B, distance="euclidean", numerics="direct"):
    """Batched arg-max similarity.  A [l, d, m], B [l, d, n] -> (vals [l, m]
    f32, inds [l, m] i64), i.e. MaxSimCuda(A, B, dim=2, mode="tn")
    (kernels/MaxSimCuda.py:184-238,296-340).

    numerics="direct": the CUDA kernel's arithmetic -- for k ascending
    ``acc = fmaf(-(a-b), (a-b), acc)`` (max_sim.cu:78-98), dot:
    ``acc = fmaf(a, b, acc)`` (:60-75).
    numerics="expanded": ``2 a.b - |a|^2 - |b|^2`` with every dot an
    ascending-k fmaf chain (what an fp32 MFMA computes; formula of
    MultiKMeans.euc_sim, clustering/MultiKMeans.py:184-209).
    Ties -> smallest centroid index (the reference's cross-block arg-max is a
    benign race, max_sim.cu:152-180).
    """
    A = np.asarray(A, dtype=F32)
    B = np.asarray(B, dtype=F32)
    l, d, m = A.shape
    n = B.shape[2]
    vals = np.empty((l, m), dtype=F32)
    inds = np.empty((l, m), dtype=np.int64)
    chunk = max(1, min(m, (1 << 24) // max(n, 1)))
    for b in range(l):
        for s in range(0, m, chunk):
            a = A[b, :, s:s + chunk]  # [d, c]
            acc = np.zeros((a.shape[1], n), dtype=F32)
            if distance == "euclidean" and numerics == "direct":
                for k in range(d):
                    dif = (a[k][:, None] - B[b, k][None, :]).astype(F32)
                    acc = _fma(-dif, dif, acc)
            elif distance in ("inner", "cosine") or numerics == "expanded":
                for k in range(d):
                    acc = _fma(a[k][:, None], B[b, k][None, :], acc)
                if distance == "euclidean":
                    a2 = np.zeros(a.shape[1], dtype=F32)
                    b2 = np.zeros(n, dtype=F32)
                    for k in range(d):
                        a2 = _fma(a[k], a[k], a2)
                        b2 = _fma(B[b, k], B[b, k], b2)
                    acc = (F32(2) * acc).astype(F32)
                    acc = (acc - a2[:, None]).astype(F32)
                    acc = (acc - b2[None, :]).astype(F32)
            else:
                raise ValueError((distance, numerics))
            inds[b, s:s + chunk] = np.argmax(acc, axis=1)  # first max = smallest index
            vals[b, s:s + chunk] = acc.max(axis=1)
    return vals, inds


def _fma(a, b, c):
    """Single-rounding fp32 fused multiply-add emulated in fp64.

    The product of two fp32 is exact in fp64 (48-bit significand); adding an
    fp32 addend in fp64 can round once (53 bits) before the final rounding to
    fp32.  That double rounding differs from a true fmaf only when the fp64
    sum sits exactly on an fp32 rounding boundary after losing bits beyond 53
    -- it needs |exponent gap| > 29 between product and addend with a tie
    pattern, which the tests' value ranges do not produce; the C oracle
    (oracle/ivfpq_oracle.c) uses the real ``fmaf``.
    """
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F32)


def compute_centroids(data, labels, k):
    """Per-cluster mean; empty cluster -> 0.

    data [l, d, n] f32, labels [l, n] i64 -> [l, d, k] f32.  Restates
    compute_centroids.cu:10-86 (sum / count, ``count == 0 ? 0 : sum/count``
    :82).  The reference accumulates with shared-memory atomics in
    non-deterministic order; here the sum is taken in fp64 and rounded once,
    so comparisons use an fp32 tolerance.
    """
    l, d, n = data.shape
    out = np.zeros((l, d, k), dtype=F32)
    for b in range(l):
        cnt = np.bincount(labels[b], minlength=k).astype(np.float64)
        for e in range(d):
            s = np.bincount(labels[b], weights=data[b, e].astype(np.float64), minlength=k)
            with np.errstate(divide="ignore", invalid="ignore"):
                mean = np.where(cnt > 0, s / np.maximum(cnt, 1), 0.0)
            out[b, e] = mean.astype(F32)
    return out


def kmeans_fit(data, centroids, max_iter, tol, distance="euclidean", numerics="direct"):
    """Lloyd driver of MultiKMeans.fit (clustering/MultiKMeans.py:415-453) for
    given initial centroids (n_redo=1): assign, update, ``error = sum((c-c')^2)``,
    stop when ``error <= tol``.  Returns (centroids, labels, n_iter)."""
    k = centroids.shape[2]
    labels = None
    it = 0
    for it in range(1, max_iter + 1):
        _, labels = max_sim(data, centroids, distance, numerics)
        new_c = compute_centroids(data, labels, k)
        err = ((centroids.astype(F32) - new_c) ** 2).sum(dtype=F32)
        centroids = new_c
        if err <= tol:
            break
    return centroids, labels, it


def kmeans_fit_redo(data, centroids, n_redo, max_iter, tol, n_clusters, distance="euclidean",
                    numerics="direct", assign=None):
    """The whole of MultiKMeans.fit (clustering/MultiKMeans.py:415-453) including the redo loop:
    redo 0 starts from ``centroids`` when given, every later redo from
    ``initialize_centroids`` (:277-283: ONE ``np.random.choice(n, [k], replace=False)`` index
    set shared by all sub-problems -- the caller seeds np.random); the redo with the smallest
    inertia ``mean(-maxsims)`` of its LAST assign wins (:440-445, strict ``<``); the labels
    returned are those of that last assign (taken BEFORE the final update).
    ``assign`` overrides the arg-max routine (e.g. the C oracle's).
    Returns (centroids, labels, inertia per redo, steps per redo)."""
    assign = assign or (lambda a, b: max_sim(a, b, distance, numerics))
    data = np.asarray(data, dtype=F32)
    n = data.shape[2]
    best = None
    inertias, steps = [], []
    for _ in range(n_redo):
        if centroids is None:
            index = np.random.choice(n, size=[n_clusters], replace=False)
            centroids = data[:, :, index].copy()
        it = 0
        for it in range(1, max_iter + 1):
            maxsims, labels = assign(data, centroids)
            new_c = compute_centroids(data, labels, n_clusters)
            err = ((centroids.astype(F32) - new_c) ** 2).sum(dtype=F32)
            centroids = new_c
            if err <= tol:
                break
        inertia = float((-maxsims).astype(F32).mean(dtype=np.float64))
        inertias.append(inertia)
        steps.append(it)
        if best is None or inertia < best[0]:
            best = (inertia, centroids, labels)
        centroids = None
    return best[1], best[2], inertias, steps


def pq_decode(codebook, codes):
    """codes u8 [m, n] -> reconstruction f32 [m*ds, n].
    Restates pq_decode.cu:8-53 / PQCodec._decode_cpu (codec/PQCodec.py:95-111)."""
    m, ds, k = codebook.shape
    n = codes.shape[1]
    out = np.empty((m, ds, n), dtype=F32)
    for j in range(m):
        out[j] = codebook[j][:, codes[j]]
    return out.reshape(m * ds, n)


# --------------------------------------------------------------------------
# container placement (a-12, a-14)
# --------------------------------------------------------------------------
def get_ioa(cells):
    """Index of appearance: rank of each element among equal labels, in input
    order.  Restates get_ioa.cu:9-47 / CellContainer._get_ioa_cpu
    (container/CellContainer.py:118-126)."""
    cells = np.asarray(cells, dtype=np.int64)
    ioa = np.empty_like(cells)
    seen = {}
    for i, c in enumerate(cells.tolist()):
        r = seen.get(c, 0)
        ioa[i] = r
        seen[c] = r + 1
    return ioa


def get_write_address(is_empty, cell_start, cell_capacity, cells, ioa):
    """The ``ioa``-th empty slot inside the cell's capacity range, -1 if none.
    Restates get_write_address_v2.cu:9-41."""
    n_slots = is_empty.shape[0]
    out = np.full(cells.shape, -1, dtype=np.int64)
    for i, (c, r) in enumerate(zip(cells.tolist(), ioa.tolist())):
        st = int(cell_start[c])
        cap = int(cell_capacity[c])
        cnt = 0
        for a in range(st, min(st + cap, n_slots)):
            if is_empty[a] == 1:
                if cnt == r:
                    out[i] = a
                    break
                cnt += 1
    return out


def get_cell_by_address(address, cell_start, cell_capacity):
    """address -> cell index, -1 when outside every [start, start+capacity).
    Restates get_div_by_address_v2.cu:9-95 / CellContainer._get_cell_by_address_cpu
    (container/CellContainer.py:97-106)."""
    end = cell_start + cell_capacity
    out = np.full(address.shape, -1, dtype=np.int64)
    for i, a in enumerate(address.tolist()):
        hit = np.nonzero((cell_start <= a) & (a < end))[0]
        if hit.size:
            out[i] = hit[0]
    return out


class ContainerState:
    """Plain-numpy mirror of CellContainer's buffers (container/CellContainer.py:46-80,
    BaseContainer.py:32-38)."""

    def __init__(self, code_size, n_cells, initial_size, expand_step_size=128,
                 expand_mode="double", contiguous_size=4):
        self.code_size = code_size
        self.n_cells = n_cells
        self.cs = contiguous_size
        self.expand_step_size = expand_step_size
        self.expand_mode = expand_mode
        cap = n_cells * initial_size
        self.storage = np.zeros((code_size // contiguous_size, cap, contiguous_size), np.uint8)
        self.cell_start = np.arange(n_cells, dtype=np.int64) * initial_size
        self.cell_size = np.zeros(n_cells, np.int64)
        self.cell_capacity = np.full(n_cells, initial_size, np.int64)
        self.is_empty = np.ones(cap, np.uint8)
        self.address2id = np.full(cap, -1, np.int64)
        self.max_id = -1

    def expand(self, cells):
        """CellContainer.expand (container/CellContainer.py:249-311)."""
        for c in cells.tolist():
            st = int(self.cell_start[c])
            cap = int(self.cell_capacity[c])
            end = st + cap
            n_new = self.expand_step_size if self.expand_mode == "step" else cap
            g, _, cs = self.storage.shape
            self.storage = np.concatenate(
                [self.storage[:, :end], np.zeros((g, n_new, cs), np.uint8), self.storage[:, end:]], axis=1)
            self.address2id = np.concatenate(
                [self.address2id[:end], np.full(n_new, -1, np.int64), self.address2id[end:]])
            self.is_empty = np.concatenate(
                [self.is_empty[:end], np.ones(n_new, np.uint8), self.is_empty[end:]])
            self.cell_capacity[c] += n_new
            self.cell_start[c + 1:] += n_new

    def add(self, codes, cells, ids=None):
        """CellContainer.add (container/CellContainer.py:313-367)."""
        n = codes.shape[1]
        if ids is None:
            ids = np.arange(n, dtype=np.int64) + self.max_id + 1
        uniq, counts = np.unique(cells, return_counts=True)
        ioa = get_ioa(cells)
        while True:
            free = self.cell_capacity[cells] - self.cell_size[cells] - (ioa + 1)
            need = np.unique(cells[free < 0])
            if need.size == 0:
                break
            self.expand(need)
        wa = get_write_address(self.is_empty, self.cell_start, self.cell_capacity, cells, ioa)
        codes_to_storage(codes, wa, self.storage)
        self.address2id[wa] = ids
        if n:
            self.max_id = max(self.max_id, int(ids.max()))
        self.is_empty[wa] = 0
        self.cell_size[uniq] += counts
        return ids, wa

    def remove(self, address):
        """Tombstone by address: the *intended* behaviour of CellContainer.remove
        (container/CellContainer.py:369-393); the reference's inverted guard
        (:381-383) makes its own remove a no-op -- documented divergence."""
        cap = self.address2id.shape[0]
        address = np.unique(address[(address >= 0) & (address < cap)])
        address = address[self.is_empty[address] == 0]
        if address.size == 0:
            return 0
        self.is_empty[address] = 1
        self.address2id[address] = -1
        cells = get_cell_by_address(address, self.cell_start, self.cell_capacity)
        u, c = np.unique(cells, return_counts=True)
        self.cell_size[u] -= c
        return int(address.size)


# --------------------------------------------------------------------------
# end-to-end search (3.1)
# --------------------------------------------------------------------------
def search(x, vq_codebook, pq_codebook, storage, is_empty, cell_start, cell_size,
           address2id, k, n_probe, distance="euclidean", use_smart_probing=True,
           temperature=30.0, scan_fn=None):
    """IVFPQIndex.search (index/IVFPQIndex.py:469-524) non-residual path.
    Returns (values, ids, address, cells, n_probe_list)."""
    x = np.asarray(x, dtype=F32)
    if distance == "cosine":
        x = x / (np.sqrt((x * x).sum(axis=0, keepdims=True, dtype=F32)) + F32(1e-9))
    sims = neg_sq_l2(x, vq_codebook)
    topk_sims, cells = topk_desc(sims, n_probe)
    nq = x.shape[1]
    if use_smart_probing and n_probe > 1:
        npl = smart_probing(topk_sims, n_probe, temperature)
    else:
        npl = np.full(nq, n_probe, dtype=np.int64)
    cs = cell_start[cells]
    sz = cell_size[cells]
    lut = adc_lut(x, pq_codebook, distance)
    fn = scan_fn or scan_topk
    vals, adr = fn(storage, lut, is_empty, cs, sz, npl, k)
    ids = get_id_by_address(address2id, adr)
    return vals, ids, adr, cells, npl
