"""TEXMEX vector files (the format of SIFT1M / GIST1M, which the reference's benchmarks are quoted
on: benchmark/turing/sift1m/README.md:10-17) and the synthetic stand-ins used when the files are
not on the machine.

.fvecs / .ivecs / .bvecs: every vector is a little-endian int32 dimension followed by that many
float32 / int32 / uint8 components.
"""
from __future__ import annotations

import os

import numpy as np


def _read_vecs(path, dtype, max_vectors=None):
    itemsize = np.dtype(dtype).itemsize
    size = os.path.getsize(path)
    if size == 0:
        return np.empty((0, 0), dtype=dtype)
    with open(path, "rb") as f:
        d = int(np.frombuffer(f.read(4), dtype="<i4")[0])
    if d <= 0:
        raise ValueError(f"{path}: bad vector dimension {d}")
    rec = 4 + d * itemsize
    if size % rec:
        raise ValueError(f"{path}: size {size} is not a multiple of the record size {rec} (d={d})")
    n = size // rec
    if max_vectors is not None:
        n = min(n, int(max_vectors))
    raw = np.fromfile(path, dtype=np.uint8, count=n * rec).reshape(n, rec)
    dims = raw[:, :4].copy().view("<i4")[:, 0]
    if not np.all(dims == d):
        raise ValueError(f"{path}: vectors of different dimensions")
    return raw[:, 4:].copy().view(np.dtype(dtype).newbyteorder("<")).reshape(n, d)


def read_fvecs(path, max_vectors=None):
    """-> float32 [n, d]"""
    return _read_vecs(path, np.float32, max_vectors)


def read_ivecs(path, max_vectors=None):
    """-> int32 [n, d] (ground-truth neighbour lists)"""
    return _read_vecs(path, np.int32, max_vectors)


def read_bvecs(path, max_vectors=None):
    """-> uint8 [n, d]"""
    return _read_vecs(path, np.uint8, max_vectors)


def write_fvecs(path, x):
    x = np.ascontiguousarray(x, dtype="<f4")
    n, d = x.shape
    rec = np.empty((n, 4 + 4 * d), dtype=np.uint8)
    rec[:, :4] = np.full((n, 1), d, dtype="<i4").view(np.uint8)
    rec[:, 4:] = x.view(np.uint8).reshape(n, 4 * d)
    rec.tofile(path)


def write_ivecs(path, x):
    x = np.ascontiguousarray(x, dtype="<i4")
    n, d = x.shape
    rec = np.empty((n, 4 + 4 * d), dtype=np.uint8)
    rec[:, :4] = np.full((n, 1), d, dtype="<i4").view(np.uint8)
    rec[:, 4:] = x.view(np.uint8).reshape(n, 4 * d)
    rec.tofile(path)


def find_texmex(data_dir, name):
    """Paths of <name>_{base,learn,query}.fvecs and <name>_groundtruth.ivecs under data_dir (also
    data_dir/<name>/), or None when the base or query file is missing."""
    if not data_dir:
        return None
    for root in (data_dir, os.path.join(data_dir, name)):
        p = {k: os.path.join(root, f"{name}_{k}.{'ivecs' if k == 'groundtruth' else 'fvecs'}")
             for k in ("base", "learn", "query", "groundtruth")}
        if os.path.exists(p["base"]) and os.path.exists(p["query"]):
            return {k: (v if os.path.exists(v) else None) for k, v in p.items()}
    return None
