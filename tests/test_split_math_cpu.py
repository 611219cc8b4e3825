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


# ---- round 3: the fp16 cascade (csrc/lloyd.hip) and the rank merge (scan_device.h) ----------
def test_level_1_dropped_piece_bound_holds_with_the_measured_norms():
    """|sum_k (a_k C_k - ah_k Ch_k)| <= |a - ah| (|Ch| + |C - Ch|) + |a| |C - Ch| for fp16 hi pieces (what emit()
    and gdecide_kernel bound level 1 with), and it is well below the worst case 2^-11 (|a| + |c|)^2"""
    rng = np.random.default_rng(7)
    worst = []
    for d in (16, 64, 128, 960):
        a = (rng.standard_normal((200, d)) * rng.uniform(0.1, 4000, (200, 1))).astype(np.float32)
        c = (rng.standard_normal((200, d)) * rng.uniform(0.1, 4000, (200, 1))).astype(np.float32)
        C = (2 * c).astype(np.float32)
        ah, Ch = a.astype(np.float16).astype(np.float32), C.astype(np.float16).astype(np.float32)
        ra, rC = (a - ah).astype(np.float32), (C - Ch).astype(np.float32)
        assert np.array_equal(ra.astype(np.float64), a.astype(np.float64) - ah), "a - ah is exact in fp32"
        a64, C64, ah64, Ch64 = (v.astype(np.float64) for v in (a, C, ah, Ch))
        dropped = np.abs((a64 * C64).sum(1) - (ah64 * Ch64).sum(1))
        n = np.linalg.norm
        bound = n(ra.astype(np.float64), axis=1) * (n(Ch64, axis=1) + n(rC.astype(np.float64), axis=1)) \
            + n(a64, axis=1) * n(rC.astype(np.float64), axis=1)
        assert (dropped <= bound * (1 + 1e-12)).all()
        assert (n(Ch64, axis=1) <= (1 + 2.0 ** -11) * n(C64, axis=1)).all()
        worst_case = 2.0 ** -11 * (n(a64, axis=1) + n(c.astype(np.float64), axis=1)) ** 2
        worst.append(float(np.median(bound / worst_case)))
    assert max(worst) < 0.6, worst   # "about 0.4 of the worst case"


def test_bound_norms_travel_as_bf16_rounded_up():
    """pack_bound_norms / unpack_bound_norms: each half is >= the fp32 value and within 2^-7 of it"""
    rng = np.random.default_rng(8)
    x = np.abs(np.concatenate([rng.standard_normal(1000) * 10.0 ** rng.uniform(-30, 30, 1000), [0.0, 1.0, 3.0e38]]))
    x = x.astype(np.float32)
    up = (((x.view(np.uint32).astype(np.uint64) + 0xFFFF) >> 16) << 16).astype(np.uint32).view(np.float32)
    finite = np.isfinite(up)
    assert (up[finite] >= x[finite]).all() and (up[finite] <= x[finite] * (1 + 2.0 ** -7)).all()
    assert np.isinf(up[~finite]).all()   # an overflow only ever widens the bound (the point is listed)


def test_rank_merge_places_every_entry_once_with_duplicates_ranked_by_list():
    """scan_device.h rank_merge: position = own position + per other list the entries that precede it (strictly
    better, or equal in an earlier list) -- a permutation of the merged order even with equal keys"""
    rng = np.random.default_rng(9)
    for L, LEN in ((8, 64), (4, 128), (3, 64)):
        pool = rng.integers(0, 50, size=L * LEN).astype(np.uint64)       # many equal keys
        lists = [np.sort(pool[l * LEN:(l + 1) * LEN])[::-1] for l in range(L)]   # best (largest) first
        out = np.full(L * LEN, -1, np.int64)
        for l in range(L):
            for i, x in enumerate(lists[l]):
                rank = i
                for l2 in range(L):
                    if l2 == l:
                        continue
                    y = lists[l2]
                    rank += int(((y > x) | ((y == x) & (l2 < l))).sum())
                assert out[rank] == -1, "two entries at one position"
                out[rank] = int(x)
        assert (out >= 0).all() and (np.diff(out) <= 0).all()
        assert np.array_equal(out, np.sort(pool)[::-1].astype(np.int64))
