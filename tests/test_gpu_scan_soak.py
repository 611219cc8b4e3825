"""GPU: randomised soak of the packed list scan with the fused finish (scan_packed_kernel RM > 0: rank merge of the
workgroup's lists, last-arriver finish of split queries, in-kernel exact redo) against the C oracle
(ivfpq_topk.cu:822-971 restated): values, addresses and ids bit for bit on random shapes -- sub-quantizer
counts, k, probe counts, splits, ragged / empty cells, tombstones, mass ties."""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle
from oracle import ivfpq_oracle as orc
from test_gpu_kernels import _random_index, T, N

pytestmark = pytest.mark.gpu

PACKED_M = (4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64, 96, 120, 128)


@pytest.mark.parametrize("seed", range(int(os.environ.get("TPQ_SOAK_SEEDS", "8"))))  # (more: TPQ_SOAK_SEEDS=100)
def test_fused_finish_random_shapes(seed):
    import torchpq_amd.kernels as K
    rng = np.random.default_rng(1000 + seed)
    for case in range(8):
        m = int(PACKED_M[rng.integers(0, len(PACKED_M))])
        k = int(rng.choice([1, 2, 7, 10, 33, 56, 57, 64, 100, 120, 121, 200, 248, 249, 300]))
        n_cells = int(rng.integers(3, 60))
        n_probe = int(rng.integers(1, min(n_cells, 24) + 1))
        nq = int(rng.integers(1, 30))
        n_split = int(rng.choice([1, 1, 2, 3, 5, 8, 16]))
        mean = int(rng.choice([5, 60, 150, 400]))
        tomb = int(rng.choice([0, 0, 30]))
        dup = float(rng.choice([0.0, 0.0, 0.3, 0.9]))
        storage, is_empty, start, sizes, a2i = _random_index(rng, m, n_cells, mean, tomb, dup)
        lut = (rng.standard_normal((m, nq, 256)) * rng.choice([1e-3, 1.0, 100.0])).astype(np.float32)
        if rng.random() < 0.3:  # coarse LUT values: many exact ties between different codes
            lut = np.round(lut).astype(np.float32)
        cells = np.stack([rng.permutation(n_cells)[:n_probe] for _ in range(nq)])
        npl = rng.integers(0, n_probe + 1, nq).astype(np.int64)
        npl[: max(1, nq // 2)] = n_probe
        cs, sz = start[cells], sizes[cells]
        ev, ea = c_oracle.scan_topk(storage, lut, is_empty, cs, sz, npl, k)
        eid = orc.get_id_by_address(a2i, ea)
        scan = K.IVFPQTopkHip(m=m)
        st = T(storage)
        packed = K.PackCodesHip()(st)
        v, a, i = scan.topk(st, T(lut), T(is_empty), T(cs), T(sz), T(npl), n_candidates=k, packed=packed,
                            address2id=T(a2i), n_split=n_split)
        tag = (seed, case, m, k, n_cells, n_probe, nq, n_split, mean, tomb, dup)
        assert np.array_equal(N(v), ev), tag
        assert np.array_equal(N(a), ea), tag
        assert np.array_equal(N(i), eid), tag


@pytest.mark.parametrize("seed", range(int(os.environ.get("TPQ_SOAK_SEEDS_LARGE", "4"))))
def test_large_batch_route_random_shapes(seed):
    """the same against the C oracle for the large-batch route at m = 64 (16-bit selection table, four- and eight-wave
    workgroups, the split last round, scan_finish_exact_kernel): batches of 1 024 - 2 600 queries, the LUT built in the
    workgroup from query and codebook, both metrics, sub-vector lengths 1 and 2, k up to 504, integer-valued codebooks
    and queries (mass ties in the 16-bit keys AND in the exact values), tombstones, duplicated codes"""
    import torchpq_amd.kernels as K
    rng = np.random.default_rng(5000 + seed)
    m = 64
    for case in range(3):
        ds = int(rng.choice([1, 2]))
        k = int(rng.choice([1, 7, 33, 100, 120, 121, 200, 248, 249, 300, 440, 441, 504]))
        n_cells = int(rng.integers(20, 90))
        n_probe = int(rng.integers(1, 17))
        nq = int(rng.integers(1024, 2600))
        mean = int(rng.choice([5, 60, 150, 400]))
        tomb = int(rng.choice([0, 0, 30]))
        dup = float(rng.choice([0.0, 0.0, 0.3, 0.9]))
        distance = str(rng.choice(["euclidean", "inner"]))
        storage, is_empty, start, sizes, a2i = _random_index(rng, m, n_cells, mean, tomb, dup)
        scale = float(rng.choice([1e-3, 1.0, 100.0]))
        codebook = (rng.standard_normal((m, ds, 256)) * scale).astype(np.float32)
        query = (rng.standard_normal((m * ds, nq)) * scale).astype(np.float32)
        if rng.random() < 0.4:  # small integers: exact ties between different codes, in both the keys and the values
            codebook = np.round(codebook / scale * 2).astype(np.float32)
            query = np.round(query / scale * 2).astype(np.float32)
        cells = np.stack([rng.permutation(n_cells)[:n_probe] for _ in range(nq)])
        npl = rng.integers(0, n_probe + 1, nq).astype(np.int64)
        npl[: nq // 2] = n_probe
        cs, sz = start[cells], sizes[cells]
        lut = c_oracle.adc_lut(query, codebook, distance)
        ev, ea = c_oracle.scan_topk(storage, lut, is_empty, cs, sz, npl, k)
        eid = orc.get_id_by_address(a2i, ea)
        scan = K.IVFPQTopkHip(m=m)
        st = T(storage)
        packed = K.PackCodesHip()(st)
        hint = int(rng.choice([0, n_probe * mean]))
        v, a, i = scan.topk_fused(st, T(query), T(codebook), T(is_empty), T(cs), T(sz), T(npl), k, distance=distance,
                                  packed=packed, address2id=T(a2i), slots_hint=hint)
        tag = (seed, case, ds, k, n_cells, n_probe, nq, mean, tomb, dup, distance, scale, hint)
        assert np.array_equal(N(v), ev), tag
        assert np.array_equal(N(a), ea), tag
        assert np.array_equal(N(i), eid), tag
