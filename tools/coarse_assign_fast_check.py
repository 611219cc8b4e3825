#!/usr/bin/env python
"""tpq_coarse_assign (error-bounded bf16 top-2 + exact re-check) against tpq_max_sim (the bit-exact
fp32 kernel): labels must be IDENTICAL; reports both times and the share of points that went to the
exact re-check.  Shapes: the coarse assign of add() -- 1 M points x n_cells x d."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=1 << 20)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--shapes", default="128x16384,128x1024,96x4096,64x1024,32x256,128x65536")
    ap.add_argument("--kind", default="clustered", choices=["clustered", "gauss", "sift"])
    a = ap.parse_args()
    from torchpq_amd import kernels as K
    dev = "cuda:0"
    res = {}
    for shp in a.shapes.split(","):
        d, n = (int(x) for x in shp.split("x"))
        g = torch.Generator(device=dev)
        g.manual_seed(d * 7 + n)
        if a.kind == "gauss":
            A = torch.randn(d, a.m, generator=g, device=dev)
            B = A[:, torch.randperm(a.m, generator=g, device=dev)[:n]].contiguous()
        else:
            centers = torch.randn(d, 4096, generator=g, device=dev).abs() * 45.0
            pick = torch.randint(0, 4096, (a.m,), generator=g, device=dev)
            A = (centers[:, pick] + torch.randn(d, a.m, generator=g, device=dev) * 12.0)
            if a.kind == "sift":
                A = A.clamp_(0, 255).round_()
            A = A.contiguous()
            B = A[:, torch.randperm(a.m, generator=g, device=dev)[:n]].contiguous() + 0.5
        exact = K.MaxSimHip(distance="euclidean")
        fast = K.CoarseAssignHip(distance="euclidean")
        le = exact(A, B, dim=1)[1]
        lf = fast(A, B)
        same = bool(torch.equal(le, lf))
        rechecked = fast.last_rechecked()
        t_e = timeit(lambda: exact(A, B, dim=1), a.iters)
        t_f = timeit(lambda: fast(A, B), a.iters)
        res[shp] = {"labels_identical": same, "mismatches": int((le != lf).sum().item()),
                    "rechecked_share": round(rechecked / a.m, 4),
                    "exact_ms": round(t_e, 3), "fast_ms": round(t_f, 3), "speedup": round(t_e / t_f, 2),
                    "fp32_equivalent_TFLOPs": round(2.0 * a.m * n * d / t_f / 1e9, 1)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
