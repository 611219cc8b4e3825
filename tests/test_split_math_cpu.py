"""CPU: the arithmetic facts the bf16 selection / split kernels rest on, checked in numpy
(csrc/kmeans_split.hip, csrc/assign_fast.hip):
  * x = x1 + x2 + x3 exactly, with x_i = bf16 round-to-nearest-even of the running residual;
  * the two-piece residual is below 2^-16 |x| (a bf16 piece carries 8 significant bits);
  * every product of two bf16 pieces is exact in fp32;
  * dropping c2 a2 + r_c a + c r_a costs at most 3.03 * 2^-16 |a||c| per term (the eps_prod of the bound);
  * the order-preserving (value, ~index) key of the split re-check sorts like (value desc, index asc).
"""
import numpy as np


def bf16_rne(x):
    """float32 -> nearest bfloat16 (ties to even), returned as float32"""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    rounded = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return rounded.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, np.float32)
    p1 = bf16_rne(x)
    r1 = (x - p1).astype(np.float32)
    p2 = bf16_rne(r1)
    r2 = (r1 - p2).astype(np.float32)
    p3 = bf16_rne(r2)
    return p1, p2, p3, r1, r2


def _samples(rng, n):
    return np.concatenate([
        (rng.standard_normal(n) * 10).astype(np.float32),
        rng.integers(0, 256, n).astype(np.float32),
        (rng.standard_normal(n) * 1e-20).astype(np.float32),
        (rng.standard_normal(n) * 1e20).astype(np.float32),
        np.float32([0.0, 1.0, -1.0, 255.0, 1.0 + 2.0 ** -23, 3.0 * 2.0 ** -130, 16777215.0]),
    ])


def test_three_piece_split_is_exact_and_two_piece_residual_is_below_2_to_minus_16():
    rng = np.random.default_rng(0)
    x = _samples(rng, 200_000)
    p1, p2, p3, r1, r2 = split3(x)
    x64 = x.astype(np.float64)
    assert np.array_equal(p1.astype(np.float64) + p2.astype(np.float64) + p3.astype(np.float64), x64)
    # the subtractions are exact in fp32 (Sterbenz-like: the residual needs <= 16 / <= 8 bits)
    assert np.array_equal(r1.astype(np.float64), x64 - p1.astype(np.float64))
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - p2.astype(np.float64))
    nz = x != 0
    assert (np.abs(r1[nz].astype(np.float64)) <= 2.0 ** -8 * np.abs(x64[nz])).all()
    assert (np.abs(r2[nz].astype(np.float64)) <= 2.0 ** -16 * np.abs(x64[nz])).all()
    # SIFT-like integers 0..255 are a single piece
    ints = rng.integers(0, 256, 1000).astype(np.float32)
    q1, q2, q3, _, _ = split3(ints)
    assert np.array_equal(q1, ints) and not q2.any() and not q3.any()


def test_piece_products_are_exact_in_fp32_and_the_dropped_ones_fit_eps_prod():
    rng = np.random.default_rng(1)
    a = (rng.standard_normal(100_000) * 7).astype(np.float32)
    c = (rng.standard_normal(100_000) * 3).astype(np.float32)
    a1, a2, a3, _, ra = split3(a)   # ra = a - a1 - a2
    c1, c2, c3, _, rc = split3(c)
    for x, y in ((a1, c1), (a1, c2), (a2, c1), (a2, c2), (a1, c3), (a3, c1)):
        exact = x.astype(np.float64) * y.astype(np.float64)
        assert np.array_equal((x * y).astype(np.float64), exact)  # 8 x 8 significant bits fit 24
    full = a.astype(np.float64) * c.astype(np.float64)
    kept2 = (a1.astype(np.float64) * c1 + a1.astype(np.float64) * c2 + a2.astype(np.float64) * c1)
    assert (np.abs(full - kept2) <= 3.03 * 2.0 ** -16 * np.abs(full) + 1e-300).all()
    kept6 = kept2 + a2.astype(np.float64) * c2 + a1.astype(np.float64) * c3 + a3.astype(np.float64) * c1
    assert (np.abs(full - kept6) <= 1.01 * 2.0 ** -23 * np.abs(full) + 1e-300).all()


def test_recheck_key_orders_like_value_desc_then_index_asc():
    rng = np.random.default_rng(2)
    v = np.concatenate([(rng.standard_normal(5000) * 100).astype(np.float32),
                        np.float32([0.0, -0.0, np.inf, -np.inf, 1e-45, -1e-45])])
    v = np.concatenate([v, v[:500]])  # exact ties
    idx = rng.permutation(v.size).astype(np.uint32)
    fb = v.view(np.uint32)
    ordered = np.where(fb & 0x80000000, ~fb, fb | np.uint32(0x80000000)).astype(np.uint64)
    key = (ordered << np.uint64(32)) | (np.uint64(0xFFFFFFFF) - idx.astype(np.uint64))
    by_key = np.argsort(key)[::-1]
    # reference order: value descending (with -0.0 < +0.0, as the bit pattern orders them), index ascending
    ref = sorted(range(v.size), key=lambda i: (-float(v[i]) if v[i] != 0 else (0.0 if not np.signbit(v[i]) else 1e-300),
                                               int(idx[i])))
    assert [int(i) for i in by_key] == ref
    # decoding returns the value bits and the index
    back = np.where(ordered & 0x80000000, ordered & 0x7FFFFFFF, ~ordered & 0xFFFFFFFF).astype(np.uint32).view(np.float32)
    assert np.array_equal(back.view(np.uint32), fb)
    assert np.array_equal((np.uint64(0xFFFFFFFF) - (key & np.uint64(0xFFFFFFFF))).astype(np.uint32), idx)
