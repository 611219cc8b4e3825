#!/usr/bin/env python
"""Randomised soak of the selection kernels: labels must equal the bit-exact fp32 kernel's on
every case.  Shapes and data kinds are drawn at random (Gaussian, SIFT-like integers, heavy-tailed,
tight clusters, large common offset, tiny and huge magnitudes, duplicated centroids).

    python tools/selection_soak.py [--cases 60] [--seed 0]           tpq_coarse_assign (d <= 128) + tpq_max_sim_select
    python tools/selection_soak.py --mode cascade --cases 200        the fp16 cascades, forced on every shape:
        tpq_coarse_assign narrow (candidate route from 2 chunks on) and wide (128 < d <= 1024), tpq_lloyd_step
(the cascade mode asks for the cascade on every shape: CoarseAssignHip.default_route = "cascade")"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchpq_amd import kernels as K  # noqa: E402

if "--mode" in sys.argv and sys.argv[sys.argv.index("--mode") + 1] == "cascade":
    K.CoarseAssignHip.default_route = "cascade"

KINDS = ("gauss", "sift", "heavy", "tight", "offset", "tiny", "huge", "dups")


def make(rng, kind, d, m, n, dev):
    if kind == "sift":
        A = rng.integers(0, 256, (d, m)).astype(np.float32)
    elif kind == "heavy":
        A = (rng.standard_t(2.5, (d, m)) * 5).astype(np.float32)
    elif kind == "tight":
        c = rng.standard_normal((d, 8)) * 50
        A = (c[:, rng.integers(0, 8, m)] + rng.standard_normal((d, m)) * 0.01).astype(np.float32)
    elif kind == "offset":
        A = (1000.0 + rng.standard_normal((d, m))).astype(np.float32)
    elif kind == "tiny":
        A = (rng.standard_normal((d, m)) * 1e-18).astype(np.float32)
    elif kind == "huge":
        A = (rng.standard_normal((d, m)) * 1e14).astype(np.float32)
    else:
        A = (rng.standard_normal((d, m)) * 7).astype(np.float32)
    B = A[:, rng.integers(0, m, n)].copy()
    if kind != "dups":
        B = B + (np.abs(A).mean() * 0.05 * rng.standard_normal((d, n))).astype(np.float32)
    return torch.from_numpy(np.ascontiguousarray(A)).to(dev), torch.from_numpy(np.ascontiguousarray(B)).to(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--mode", default="selection", choices=["selection", "cascade", "probe"])
    ap.add_argument("--trace", action="store_true", help="print every case before it runs and synchronise after it")
    ap.add_argument("--only-case", type=int, default=-1, help="cascade mode: run this case alone (same random stream)")
    ap.add_argument("--big", action="store_true", help="cascade mode: up to 200 000 points x 20 000 centroids; probe mode: up to 19 168 cells")
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    dev = "cuda:0"
    if a.mode == "cascade":
        return cascade_soak(a, rng, dev)
    if a.mode == "probe":
        return probe_soak(a, rng, dev)
    bad, log = [], {"coarse": 0, "select": 0, "rechecked_share_max": 0.0}
    for c in range(a.cases):
        kind = KINDS[c % len(KINDS)]
        d = int(rng.integers(1, 129))
        m = int(rng.integers(1, 20000))
        n = int(rng.integers(1, 3000))
        dist = "euclidean" if rng.random() < 0.75 else "inner"
        A, B = make(rng, kind, d, m, n, dev)
        op = K.CoarseAssignHip(distance=dist)
        got = op(A, B)
        want = K.MaxSimHip(distance=dist)(A, B, dim=1)[1]
        log["coarse"] += 1
        log["rechecked_share_max"] = max(log["rechecked_share_max"], op.last_rechecked() / m)
        if not torch.equal(got, want):
            bad.append(("coarse", kind, d, m, n, dist, int((got != want).sum())))
        l = int(rng.integers(1, 5))
        d2 = int(rng.integers(1, 65))
        n2 = int(rng.integers(1, 257))
        As, Bs = zip(*[make(rng, kind, d2, m, n2, dev) for _ in range(l)])
        A3, B3 = torch.stack(As).contiguous(), torch.stack(Bs).contiguous()
        got = K.MaxSimSelectHip(distance=dist)(A3, B3)[1]
        want = K.MaxSimHip(distance=dist)(A3, B3, dim=2)[1]
        log["select"] += 1
        if not torch.equal(got, want):
            bad.append(("select", kind, l, d2, m, n2, dist, int((got != want).sum())))
    log["mismatching_cases"] = bad
    print(json.dumps(log))
    sys.exit(1 if bad else 0)


def probe_soak(a, rng, dev):
    """the coarse step of search(): fp16 selection + exact candidates (route "fp16", prepared block or not) against the
    fp32-MFMA route -- sims, cells, extents and probe counts must be equal, bit for bit"""
    bad, log = [], {"cases": 0, "skipped_unsupported": 0}
    for c in range(a.cases):
        kind = KINDS[c % len(KINDS)]
        d = int(rng.integers(1, 129))
        nq = int(rng.integers(1, 3000))
        n_cells = 32 * int(rng.integers(8, 600 if a.big else 200))
        n_probe = int(min(n_cells, rng.choice([1, 2, 7, 16, 33, 64, 100, 128, 200, 500])))
        A, B = make(rng, kind, d, nq, n_cells, dev)          # queries [d, nq], centroids [d, n_cells]
        sizes = torch.randint(0, 300, (n_cells,), device=dev)
        start = torch.cumsum(sizes + 3, 0) - sizes - 3
        smart = 30.0 if c % 2 else None
        want = K.CoarseProbeHip(route="fp32")(A, B, start, sizes, n_probe, smart)
        prep = K.CoarseProbeHip.prepare(B) if c % 3 else None
        got = K.CoarseProbeHip(route="fp16")(A, B, start, sizes, n_probe, smart, prepared=prep)
        log["cases"] += 1
        # rows whose similarities are NaN have no defined order on either route
        ok = ~torch.isnan(want[0]).any(dim=1)
        if not all(torch.equal(x[ok], y[ok]) for x, y in zip(want, got)):
            bad.append((kind, d, nq, n_cells, n_probe, smart is not None, prep is not None,
                        int((want[1][ok] != got[1][ok]).any(dim=1).sum())))
    log["mismatching_cases"] = bad
    print(json.dumps(log))
    sys.exit(1 if bad else 0)


def cascade_soak(a, rng, dev):
    bad = []
    log = {"narrow": 0, "wide": 0, "lloyd": 0, "exact_step_share_max": {"narrow": 0.0, "wide": 0.0}}
    for c in range(a.cases):
        kind = KINDS[c % len(KINDS)]
        dist = "euclidean" if rng.random() < 0.7 else "inner"
        # narrow: d <= 128; euclidean problems take the cascade (chunked from 257 centroids on: candidate route)
        mmax, nmax = (200000, 20000) if a.big else (30000, 6000)
        d, m, n = int(rng.integers(1, 129)), int(rng.integers(1, mmax)), int(rng.integers(1, nmax))
        A, B = make(rng, kind, d, m, n, dev)
        run = a.only_case < 0 or a.only_case == c
        if a.trace and run:
            print("case", c, "narrow", kind, d, m, n, dist, flush=True)
        if run:
            op = K.CoarseAssignHip(distance=dist)
            got = op(A, B)
            if a.trace:
                torch.cuda.synchronize()
            want = K.MaxSimHip(distance=dist)(A, B, dim=1)[1]
            log["narrow"] += 1
            log["exact_step_share_max"]["narrow"] = max(log["exact_step_share_max"]["narrow"], op.last_rechecked() / m)
            if not torch.equal(got, want):
                bad.append(("narrow", kind, d, m, n, dist, int((got != want).sum())))
        # wide: 128 < d <= 1024, both metrics
        d, m, n = int(rng.integers(129, 1025)), int(rng.integers(1, 60000 if a.big else 12000)), \
            int(rng.integers(1, 8000 if a.big else 3000))
        A, B = make(rng, kind, d, m, n, dev)
        if a.trace and run:
            print("case", c, "wide", kind, d, m, n, dist, flush=True)
        if run:
            op = K.CoarseAssignHip(distance=dist)
            got = op(A, B)
            if a.trace:
                torch.cuda.synchronize()
                print("  exact-step share", op.last_rechecked() / m, flush=True)
            want = K.MaxSimHip(distance=dist)(A, B, dim=1)[1]
            log["wide"] += 1
            log["exact_step_share_max"]["wide"] = max(log["exact_step_share_max"]["wide"], op.last_rechecked() / m)
            if not torch.equal(got, want):
                bad.append(("wide", kind, d, m, n, dist, int((got != want).sum())))
        # one Lloyd step on prepared data (euclidean, d <= 64, k <= 256)
        l, d2, n2 = int(rng.integers(1, 5)), int(rng.integers(1, 65)), int(rng.integers(1, 257))
        As, Bs = zip(*[make(rng, kind, d2, m, n2, dev) for _ in range(l)])
        A3, B3 = torch.stack(As).contiguous(), torch.stack(Bs).contiguous()
        if a.trace and run:
            print("case", c, "lloyd", kind, l, d2, m, n2, flush=True)
        if run and K.LloydStepHip.supported(l, d2, m, n2):
            step = K.LloydStepHip(A3, B3)
            _, got, _ = step(B3, update=False)
            want = K.MaxSimHip(distance="euclidean")(A3, B3, dim=2)[1]
            log["lloyd"] += 1
            if not torch.equal(got, want):
                bad.append(("lloyd", kind, l, d2, m, n2, int((got != want).sum())))
    log["mismatching_cases"] = bad
    print(json.dumps(log))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
