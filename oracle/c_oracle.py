"""ctypes binding of oracle/ivfpq_oracle.c (TEST INFRASTRUCTURE ONLY).

Used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
Never imported from torchpq_amd/.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libivfpq_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ivfpq_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libivfpq_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.oracle_scan_topk.restype = C.c_int64
        _lib.oracle_scan_topk_residual.restype = C.c_int64
        _lib.oracle_adc_lut.restype = None
        _lib.oracle_max_sim.restype = None
        _lib.oracle_coarse_sims.restype = None
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def scan_topk(storage, lut, is_empty, cell_start, cell_size, n_probe_list, k,
              n_threads=None, return_scanned=False):
    storage = np.ascontiguousarray(storage, dtype=np.uint8)
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    cell_start = np.ascontiguousarray(cell_start, dtype=np.int64)
    cell_size = np.ascontiguousarray(cell_size, dtype=np.int64)
    n_probe_list = np.ascontiguousarray(n_probe_list, dtype=np.int64)
    g, n_slots, cs = storage.shape
    assert cs == 4
    m = g * cs
    nq, max_np = cell_start.shape
    assert lut.shape == (m, nq, 256)
    vals = np.empty((nq, k), np.float32)
    adr = np.empty((nq, k), np.int64)
    ie = None
    if is_empty is not None:
        ie = np.ascontiguousarray(is_empty, dtype=np.uint8)
    nt = n_threads or os.cpu_count() or 1
    scanned = lib().oracle_scan_topk(
        _p(storage), _p(lut), _p(ie) if ie is not None else None, _p(cell_start), _p(cell_size),
        _p(n_probe_list), _p(vals), _p(adr), C.c_int64(n_slots), C.c_int(nq), C.c_int(max_np),
        C.c_int(m), C.c_int(k), C.c_int(nt))
    if return_scanned:
        return vals, adr, int(scanned)
    return vals, adr


def scan_topk_residual(storage, part1, part2, cells, base_sims, is_empty, cell_start, cell_size,
                       n_probe_list, k, full=None, n_threads=None):
    storage = np.ascontiguousarray(storage, dtype=np.uint8)
    g, n_slots, cs = storage.shape
    m = g * cs
    cell_start = np.ascontiguousarray(cell_start, dtype=np.int64)
    cell_size = np.ascontiguousarray(cell_size, dtype=np.int64)
    n_probe_list = np.ascontiguousarray(n_probe_list, dtype=np.int64)
    base_sims = np.ascontiguousarray(base_sims, dtype=np.float32)
    nq, max_np = cell_start.shape
    p1 = p2 = fl = cl = None
    if full is not None:
        fl = np.ascontiguousarray(full, dtype=np.float32)
        assert fl.shape == (nq, max_np, m, 256)
    else:
        p1 = np.ascontiguousarray(part1, dtype=np.float32)
        p2 = np.ascontiguousarray(part2, dtype=np.float32)
        cl = np.ascontiguousarray(cells, dtype=np.int64)
        assert p1.shape == (nq, m, 256) and p2.shape[1:] == (m, 256)
    ie = np.ascontiguousarray(is_empty, dtype=np.uint8) if is_empty is not None else None
    vals = np.empty((nq, k), np.float32)
    adr = np.empty((nq, k), np.int64)
    nt = n_threads or os.cpu_count() or 1
    P = lambda a: _p(a) if a is not None else None
    lib().oracle_scan_topk_residual(
        _p(storage), P(p1), P(p2), P(fl), P(cl), _p(base_sims), P(ie), _p(cell_start), _p(cell_size),
        _p(n_probe_list), _p(vals), _p(adr), C.c_int64(n_slots), C.c_int(nq), C.c_int(max_np),
        C.c_int(m), C.c_int(k), C.c_int(nt))
    return vals, adr


def adc_lut(query, codebook, distance="euclidean", n_threads=None):
    query = np.ascontiguousarray(query, dtype=np.float32)
    codebook = np.ascontiguousarray(codebook, dtype=np.float32)
    m, ds, k = codebook.shape
    assert k == 256
    d, nq = query.shape
    assert d == m * ds
    lut = np.empty((m, nq, 256), np.float32)
    nt = n_threads or os.cpu_count() or 1
    lib().oracle_adc_lut(_p(query), _p(codebook), _p(lut), C.c_int(m), C.c_int(ds), C.c_int(nq),
                         C.c_int(1 if distance == "euclidean" else 0), C.c_int(nt))
    return lut


_MODES = {("euclidean", "direct"): 0, ("inner", "direct"): 1, ("cosine", "direct"): 1,
          ("inner", "expanded"): 1, ("cosine", "expanded"): 1, ("euclidean", "expanded"): 2}


def max_sim(A, B, distance="euclidean", numerics="direct", n_threads=None):
    A = np.ascontiguousarray(A, dtype=np.float32)
    B = np.ascontiguousarray(B, dtype=np.float32)
    l, d, m = A.shape
    n = B.shape[2]
    vals = np.empty((l, m), np.float32)
    inds = np.empty((l, m), np.int64)
    nt = n_threads or os.cpu_count() or 1
    lib().oracle_max_sim(_p(A), _p(B), _p(vals), _p(inds), C.c_int(l), C.c_int(d), C.c_int(m),
                         C.c_int(n), C.c_int(_MODES[(distance, numerics)]), C.c_int(nt))
    return vals, inds


def coarse_sims(x, centroids, n_threads=None):
    """sims [nq, n_cells] of the coarse step in the fp32-MFMA kernels' arithmetic (bit-exact check of
    tpq_ivfpq_coarse_probe); x [d, nq], centroids [d, n_cells]"""
    x = np.ascontiguousarray(x, dtype=np.float32)
    centroids = np.ascontiguousarray(centroids, dtype=np.float32)
    d, nq = x.shape
    assert centroids.shape[0] == d
    n_cells = centroids.shape[1]
    sims = np.empty((nq, n_cells), np.float32)
    nt = n_threads or os.cpu_count() or 1
    lib().oracle_coarse_sims(_p(x), _p(centroids), _p(sims), C.c_int(d), C.c_int(nq), C.c_int(n_cells),
                             C.c_int(nt))
    return sims
