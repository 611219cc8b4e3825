#!/usr/bin/env python
"""tpq_coarse_assign on wide vectors (128 < d <= 1024, the GEMM-shaped cascade of lloyd.hip) against the
fp32 kernel: labels (must be equal), share of points pass 1 leaves undecided, candidate pairs per such point, time.

    python tools/wide_check.py [--shapes d,m,n;d,m,n...] [--data gauss|clustered] [--metric euclidean|inner]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="130,1000,300;960,5000,1000;200,70000,2048")
    ap.add_argument("--data", default="gauss")
    ap.add_argument("--metric", default="euclidean")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--no-fp32-time", action="store_true")
    args = ap.parse_args()
    from torchpq_amd import kernels as K
    dev = "cuda:0"
    g = torch.Generator(device=dev)
    g.manual_seed(4321)

    def timeit(fn, iters=args.iters):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters

    for shp in args.shapes.split(";"):
        d, m, n = (int(x) for x in shp.split(","))
        if args.data == "gauss":
            A = torch.randn(d, m, generator=g, device=dev)
        else:  # clusters around positive means (uncentred, like descriptors)
            cen = torch.randn(d, 256, generator=g, device=dev).abs() * 30
            a = torch.randint(0, 256, (m,), generator=g, device=dev)
            A = (cen[:, a] + torch.randn(d, m, generator=g, device=dev) * 12).abs()
        B = A[:, torch.randperm(m, generator=g, device=dev)[:n]].contiguous()
        if n > m:
            B = torch.randn(d, n, generator=g, device=dev)
        op = K.CoarseAssignHip(distance=args.metric)
        vals, lab = op(A, B, return_vals=True)
        torch.cuda.synchronize()
        ws = op._last[0]
        off = 256 + (max(d, 128) * 4 + 255) // 256 * 256 + 28
        c1, cfb, npairs, oflag = (int(x) for x in ws[off:off + 16].view(torch.int32).tolist())
        v32, l32 = K.MaxSimHip(distance=args.metric)(A[None], B[None], dim=2, mode="tn")
        out = {"shape": [d, m, n], "data": args.data, "metric": args.metric,
               "labels_equal_fp32": float((lab == l32[0]).double().mean().item()),
               "n_diff": int((lab != l32[0]).sum().item()),
               "vals_max_rel_err": float(((vals - v32[0]).abs().max() / v32[0].abs().max()).item()),
               "undecided": round(c1 / m, 5), "pairs_per_undecided": round(npairs / max(c1, 1), 3),
               "fallback_points": cfb, "pair_overflow": oflag,
               "cascade_ms": round(timeit(lambda: op(A, B)), 3)}
        if not args.no_fp32_time:
            out["fp32_ms"] = round(timeit(lambda: K.MaxSimHip(distance=args.metric)(A[None], B[None], dim=2, mode="tn"), 1), 3)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
