"""GPU, round 4: the boundary contract and the untested branches VERDICT r3 named.

* the tickets of the one-launch scan finish are the CALLER's (tpq_ivfpq_*_tickets, SURVEY 8b "Ownership";
  torchpq/kernels/IVFPQTopkCuda.py:81-142 allocates nothing but its outputs either): a split query finished on
  a caller-owned buffer equals the oracle, leaves the buffer zero, and the entry points without tickets give
  the same rows through three launches;
* a captured graph consumes its overflow flags: a replay that needed the exact redo does not send the next
  replay through it;
* the coarse-step branches of IVFPQIndex.probe (torchpq/index/IVFPQIndex.py:485-497): use_cublas=False
  (KMeans.topk, clustering/KMeans.py:449-480), use_fused_probe=False, n_probe > 1024;
* sub-quantizer counts beyond the instantiated scan-layout kernels: m = 132 and 152 (util.max_subvectors).
"""
import ctypes

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


@pytest.fixture(scope="module")
def K():
    import torchpq_amd.kernels as k
    from torchpq_amd import _lib
    _lib.load()
    return k


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


# ---------------------------------------------------------------------------------------------
# caller-owned tickets
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m,k,n_split", [(64, 100, 16), (16, 10, 7), (32, 248, 2), (128, 100, 5)])
def test_split_queries_finish_on_caller_owned_tickets(K, m, k, n_split):
    from test_gpu_kernels import _random_index
    from torchpq_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(m * 7 + k)
    n_cells, nq, n_probe = 40, 11, 12
    storage, is_empty, start, sizes, a2i = _random_index(rng, m, n_cells, 400)
    lut = (rng.standard_normal((m, nq, 256)) * 100).astype(np.float32)
    cells = np.stack([rng.permutation(n_cells)[:n_probe] for _ in range(nq)])
    npl = np.full(nq, n_probe, np.int64)
    cs, sz = start[cells], sizes[cells]
    ev, ea = c_oracle.scan_topk(storage, lut, is_empty, cs, sz, npl, k)
    st, emp, lt = T(storage), T(is_empty), T(lut)
    tcs, tsz, tnpl = T(cs), T(sz), T(npl)
    packed = K.PackCodesHip()(st)
    ws_bytes = lib.tpq_ivfpq_scan_workspace_bytes(nq, k, n_split, m)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.tpq_ivfpq_scan_tickets_bytes(nq) == 4 * nq

    def call(tickets):
        ws = torch.empty(ws_bytes, device=DEV, dtype=torch.uint8).random_()   # garbage, as a caller's would be
        v = torch.empty(nq, k, device=DEV)
        a = torch.empty(nq, k, device=DEV, dtype=torch.int64)
        rc = lib.tpq_ivfpq_scan_topk_packed_tickets(
            _p(packed), _p(st), _p(lt), _p(emp), _p(tcs), _p(tsz), _p(tnpl), _p(v), _p(a), None, None,
            storage.shape[1], nq, n_probe, m, k, n_split, _p(ws), ws_bytes, _p(tickets), 0, stream)
        assert rc == 0, _lib.last_error()
        return v, a

    tickets = torch.zeros(nq, device=DEV, dtype=torch.int32)
    for _ in range(5):   # the same buffer, call after call: every call leaves it zero
        v, a = call(tickets)
        assert np.array_equal(N(v), ev) and np.array_equal(N(a), ea)
        assert int(tickets.abs().sum().item()) == 0
    v, a = call(None)    # no tickets: the three-launch finish, same rows
    assert np.array_equal(N(v), ev) and np.array_equal(N(a), ea)
    rc = lib.tpq_ivfpq_scan_topk_packed(
        _p(packed), _p(st), _p(lt), _p(emp), _p(tcs), _p(tsz), _p(tnpl), _p(v), _p(a), None, None,
        storage.shape[1], nq, n_probe, m, k, n_split,
        _p(torch.empty(ws_bytes, device=DEV, dtype=torch.uint8)), ws_bytes, stream)
    assert rc == 0
    torch.cuda.synchronize()
    assert np.array_equal(N(v), ev) and np.array_equal(N(a), ea)


def test_python_wrapper_keeps_one_ticket_buffer_per_stream(K):
    """IVFPQTopkHip hands every (device, stream) its own zeroed buffer; interleaved calls on two streams and a
    failing call in between leave every later result equal to the oracle's"""
    from test_gpu_kernels import _random_index
    rng = np.random.default_rng(4)
    m, k, n_split, n_cells, nq, n_probe = 64, 50, 8, 30, 6, 10
    storage, is_empty, start, sizes, _ = _random_index(rng, m, n_cells, 300)
    lut = (rng.standard_normal((m, nq, 256)) * 100).astype(np.float32)
    cells = np.stack([rng.permutation(n_cells)[:n_probe] for _ in range(nq)])
    npl = np.full(nq, n_probe, np.int64)
    cs, sz = start[cells], sizes[cells]
    ev, ea = c_oracle.scan_topk(storage, lut, is_empty, cs, sz, npl, k)
    scan = K.IVFPQTopkHip(m=m)
    st = T(storage)
    packed = K.PackCodesHip()(st)
    args = (st, T(lut), T(is_empty), T(cs), T(sz), T(npl))
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    outs = []
    torch.cuda.synchronize()
    for it in range(30):
        with torch.cuda.stream(streams[it % 2]):
            outs.append(scan.topk(*args, n_candidates=k, packed=packed, n_split=n_split))
    torch.cuda.synchronize()
    assert len(scan._ticket_cache) == 2
    for v, a in outs:
        assert np.array_equal(N(v), ev) and np.array_equal(N(a), ea)
    for t in scan._ticket_cache.values():
        assert int(t.abs().sum().item()) == 0


def test_graph_replay_consumes_its_overflow_flags(K):
    """One captured scan, replayed on an adversarial table (all of query 0's best vectors in tiles of wave 0: the
    short per-wave list overflows, the query is redone exactly) and then on a benign one: both replays return
    the oracle's rows, and the benign replay takes no redo -- the flag the first replay raised was consumed
    (the epoch that marks a raised flag is baked into the captured kernel arguments)."""
    m, k, n, nq = 16, 100, 32768, 3
    rng = np.random.default_rng(m * 1000 + k)
    lut_bad = (rng.standard_normal((m, nq, 256)) * 100).astype(np.float32)
    lut_ok = (rng.standard_normal((m, nq, 256)) * 100).astype(np.float32)
    codes = rng.integers(0, 256, (n, m), dtype=np.uint8)
    best = lut_bad[:, 0, :].argmax(axis=1).astype(np.uint8)
    top = np.nonzero((np.arange(n) // 64) % 32 == 0)[0]
    codes[top] = best
    for t in top:
        j = rng.choice(m, 2, replace=False)
        codes[t, j] = rng.integers(0, 256, 2)
    storage = np.ascontiguousarray(codes.reshape(n, m // 4, 4).transpose(1, 0, 2))
    is_empty = np.zeros(n, np.uint8)
    cs, sz, npl = np.zeros((nq, 1), np.int64), np.full((nq, 1), n, np.int64), np.ones(nq, np.int64)
    want = {name: c_oracle.scan_topk(storage, lut, is_empty, cs, sz, npl, k)
            for name, lut in (("bad", lut_bad), ("ok", lut_ok))}
    scan = K.IVFPQTopkHip(m=m)
    scan.keep_workspace = True
    st = T(storage)
    packed = K.PackCodesHip()(st)
    lut = T(lut_ok).clone()
    targs = (T(is_empty), T(cs), T(sz), T(npl))
    for n_split in (1, 2):
        scan.ticket_buffer = torch.zeros(nq, device=DEV, dtype=torch.int32)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            scan.topk(st, lut, *targs, n_candidates=k, packed=packed, n_split=n_split, slots_hint=n)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            # (slots_hint: one long cell -> the short per-wave lists this test is about)
            v, a = scan.topk(st, lut, *targs, n_candidates=k, packed=packed, n_split=n_split, slots_hint=n)
        redone = {}
        for name, src in (("bad", lut_bad), ("ok", lut_ok), ("bad", lut_bad), ("ok", lut_ok)):
            lut.copy_(T(src))
            graph.replay()
            torch.cuda.synchronize()
            ev, ea = want[name]
            assert np.array_equal(N(v), ev) and np.array_equal(N(a), ea), (n_split, name)
            redone[name] = scan.last_redone(nq)
        assert redone["bad"] >= 1 and redone["ok"] == 0, (n_split, redone)
        scan.ticket_buffer = None


# ---------------------------------------------------------------------------------------------
# the coarse-step branches of IVFPQIndex.probe, and wide codes
# ---------------------------------------------------------------------------------------------
def _build(d, m, n_cells, n, seed):
    from torchpq_amd.index import IVFPQIndex
    g = torch.Generator(device=DEV)
    g.manual_seed(seed)
    centers = torch.randn(d, 64, generator=g, device=DEV) * 4

    def sample(count):
        a = torch.randint(0, 64, (count,), generator=g, device=DEV)
        return (centers[:, a] + torch.randn(d, count, generator=g, device=DEV)).contiguous()

    np.random.seed(seed)
    idx = IVFPQIndex(d_vector=d, n_subvectors=m, n_cells=n_cells, initial_size=max(16, 2 * n // n_cells),
                     device=DEV)
    idx.train(sample(min(n, 40_000)))
    idx.add(sample(n))
    idx.use_smart_probing = False
    return idx, sample


def _check_against_oracle(idx, xq, k, exact_cells):
    """search() == the oracle's scan over the cells the index probed (bit for bit); the probed cells == the
    oracle's coarse step -- exactly on the fp32-MFMA branches, up to near-ties of a BLAS-ordered GEMM otherwise"""
    x = N(xq)
    sims, cells, npl = idx.probe(xq)
    cells_n = N(cells)
    full = c_oracle.coarse_sims(x, N(idx.vq_codec.codebook))
    osims, ocells = orc.topk_desc(full, idx.n_probe)
    if exact_cells:
        assert np.array_equal(cells_n, ocells)
    else:
        same = np.mean([len(np.intersect1d(cells_n[q], ocells[q])) / idx.n_probe for q in range(x.shape[1])])
        assert same >= 0.995, same
        scale = np.abs(osims).max()
        assert np.abs(np.sort(N(sims), 1) - np.sort(osims, 1)).max() <= 1e-4 * scale
    lut = c_oracle.adc_lut(x, N(idx.pq_codec.codebook), idx.distance)
    cs, sz = N(idx._cell_start)[cells_n], N(idx._cell_size)[cells_n]
    ev, ea = c_oracle.scan_topk(N(idx._storage), lut, N(idx._is_empty), cs, sz, N(npl), k)
    ei = orc.get_id_by_address(N(idx._address2id), ea)
    v, i = idx.search(xq, k=k)
    assert np.array_equal(N(v), ev) and np.array_equal(N(i), ei)


@pytest.mark.parametrize("branch", ["fused", "library_gemm", "no_cublas", "n_probe_1500"])
def test_every_coarse_branch_of_probe_searches_like_the_oracle(branch):
    n_cells = 2048 if branch == "n_probe_1500" else 256
    idx, sample = _build(64, 16, n_cells, 60_000, seed=5)
    xq = sample(200)
    idx.n_probe = 1500 if branch == "n_probe_1500" else 24
    if branch == "library_gemm":
        idx.use_fused_probe = False
    elif branch == "no_cublas":
        idx.use_cublas = False
    _check_against_oracle(idx, xq, k=20, exact_cells=branch == "fused")
    if branch == "n_probe_1500":   # (:380-388: beyond 1024 probes neither fused path applies)
        assert not (idx.use_fused_probe and idx.use_cublas and idx.n_probe <= 1024)


@pytest.mark.parametrize("m", [132, 152])
def test_wide_codes_beyond_the_instantiated_scan_layouts(m):
    """util.max_subvectors() admits m up to 152; there is no scan-layout kernel above 128: the reference-layout
    kernel (runtime m) serves the search, same rows as the oracle"""
    from torchpq_amd import util
    from torchpq_amd.kernels import PACKED_M
    assert m <= util.max_subvectors() and m not in PACKED_M
    idx, sample = _build(m * 2, m, 32, 20_000, seed=m)
    idx.n_probe = 6
    xq = sample(50)
    _check_against_oracle(idx, xq, k=10, exact_cells=True)
    for k in (1, 100):
        x = N(xq)
        _, cells, npl = idx.probe(xq)
        lut = c_oracle.adc_lut(x, N(idx.pq_codec.codebook), idx.distance)
        cs, sz = N(idx._cell_start)[N(cells)], N(idx._cell_size)[N(cells)]
        ev, ea = c_oracle.scan_topk(N(idx._storage), lut, N(idx._is_empty), cs, sz, N(npl), k)
        v, i = idx.search(xq, k=k)
        assert np.array_equal(N(v), ev)
        assert np.array_equal(N(i), orc.get_id_by_address(N(idx._address2id), ea))


# ---------------------------------------------------------------------------------------------
# large k (k > 248 leaves the one-launch finish; reference: fn/IVFPQTopk.py:64-104 serves k <= 1024)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("m", [4, 8, 12, 16, 24, 32, 40, 64, 96, 120, 128])
@pytest.mark.parametrize("k,n_split", [(300, 1), (500, 3), (1000, 1), (1016, 2), (249, 5), (1000, 7), (700, 4)])
def test_large_k_equals_the_oracle(K, m, k, n_split):
    """k > 248 leaves the one-launch finish: sorted lists + merge kernel up to k = 504 (768 at m > 64), pool mode above
    (csrc/scan_device.h; more splits than the ranking kernel's LDS takes are clamped inside the library)"""
    from test_gpu_kernels import _random_index
    rng = np.random.default_rng(m * 10007 + k)
    n_cells, nq, n_probe = 48, 7, 20
    storage, is_empty, start, sizes, a2i = _random_index(rng, m, n_cells, 330, n_tomb=40, dup_frac=0.02)
    lut = (rng.standard_normal((m, nq, 256)) * 100).astype(np.float32)
    cells = np.stack([rng.permutation(n_cells)[:n_probe] for _ in range(nq)])
    cells[1, 5] = cells[1, 2]              # a cell listed twice, non-adjacent: its slots count twice
    npl = np.full(nq, n_probe, np.int64)
    npl[2] = 1                             # fewer than k candidates: padded rows
    cs, sz = start[cells], sizes[cells]
    ev, ea = c_oracle.scan_topk(storage, lut, is_empty, cs, sz, npl, k)
    ei = orc.get_id_by_address(a2i, ea)
    scan = K.IVFPQTopkHip(m=m)
    st = T(storage)
    v, a, i = scan.topk(st, T(lut), T(is_empty), T(cs), T(sz), T(npl), n_candidates=k,
                        packed=K.PackCodesHip()(st), n_split=n_split, address2id=T(a2i))
    assert np.array_equal(N(v), ev) and np.array_equal(N(a), ea) and np.array_equal(N(i), ei)


@pytest.mark.parametrize("m,k", [(64, 1000), (16, 400)])
def test_large_k_on_ascending_values(K, m, k):
    """slots arranged so that every slot beats all before it (ascending values along the scan order): each one
    passes the admission threshold -- the worst case for the candidate queues and lists"""
    rng = np.random.default_rng(m + k)
    n, nq = 40000, 2
    lut = np.zeros((m, nq, 256), np.float32)
    lut[0, :, :] = np.arange(256, dtype=np.float32)[None, :] * 256.0
    lut[1, :, :] = np.arange(256, dtype=np.float32)[None, :]
    codes = np.zeros((n, m), np.uint8)
    order = np.arange(n) % 65536
    codes[:, 0] = order >> 8
    codes[:, 1] = order & 255
    storage = np.ascontiguousarray(codes.reshape(n, m // 4, 4).transpose(1, 0, 2))
    is_empty = np.zeros(n, np.uint8)
    cs, sz, npl = np.zeros((nq, 1), np.int64), np.full((nq, 1), n, np.int64), np.ones(nq, np.int64)
    ev, ea = c_oracle.scan_topk(storage, lut, is_empty, cs, sz, npl, k)
    scan = K.IVFPQTopkHip(m=m)
    st = T(storage)
    v, a = scan.topk(st, T(lut), T(is_empty), T(cs), T(sz), T(npl), n_candidates=k,
                     packed=K.PackCodesHip()(st), n_split=1)
    assert np.array_equal(N(v), ev) and np.array_equal(N(a), ea)


# ---------------------------------------------------------------------------------------------
# use_tensor_core: the coarse step selected on the fp16 matrix cores, through the index
# ---------------------------------------------------------------------------------------------
def test_use_tensor_core_changes_nothing_but_the_route():
    """IVFPQIndex.use_tensor_core (reference: index/IVFPQIndex.py:98-125, an fp16 coarse GEMM with its errors): here
    the knob forces the fp16 SELECTION pass; cells, similarities, probe counts and search results stay the fp32
    route's bit for bit -- also through a captured graph, and after the codebook is replaced (the prepared block
    follows it)"""
    idx, sample = _build(64, 16, 2048, 80_000, seed=9)
    idx.n_probe = 24
    idx.use_smart_probing = True
    xq = sample(700)
    assert idx.use_tensor_core is False
    idx._coarse_probe.route = "fp32"
    want_probe = [t.clone() for t in idx.probe(xq)]
    want = [t.clone() for t in idx.search(xq, k=10)]
    idx.use_tensor_core = True
    assert idx.use_tensor_core is True and idx._coarse_probe.route == "fp16"
    for a, b in zip(want_probe, idx.probe(xq)):
        assert torch.equal(a, b)
    got = idx.search(xq, k=10)
    assert torch.equal(want[0], got[0]) and torch.equal(want[1], got[1])
    prepared_before = idx._probe_prepared()
    assert prepared_before is not None and idx._probe_prepared() is prepared_before   # cached per codebook
    g = idx.graphed_search(700, k=10)
    gv, gi = g(xq)
    assert torch.equal(gv, want[0]) and torch.equal(gi, want[1])
    _check_against_oracle(idx, xq[:, :100].contiguous(), k=10, exact_cells=True)
    # a new codebook tensor: the prepared block is rebuilt, results follow the new codebook
    idx.vq_codec.kmeans.register_buffer("centroids", (idx.vq_codec.codebook * 1.01).contiguous())
    assert idx._probe_prepared() is not prepared_before
    idx._coarse_probe.route = "fp32"
    w2 = [t.clone() for t in idx.probe(xq)]
    idx.use_tensor_core = True
    for a, b in zip(w2, idx.probe(xq)):
        assert torch.equal(a, b)
    idx.use_tensor_core = False
    assert idx._coarse_probe.route == "auto"
