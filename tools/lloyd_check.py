#!/usr/bin/env python
"""tpq_lloyd_prepare / tpq_lloyd_step against the separate kernels at BASELINE.json configs[4]
(n_kmeans=64, d=64, n=1M, k=256): labels vs the fp32 kernel (must be equal), new centroids vs
tpq_compute_centroids, share of re-checked points, and the time of one Lloyd iteration.

    python tools/lloyd_check.py [--l 64 --d 64 --n 1000000 --k 256] [--data gauss|sift|lloyd3]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--l", type=int, default=64)
    ap.add_argument("--d", type=int, default=64)
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--k", type=int, default=256)
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--data", default="gauss")
    ap.add_argument("--lloyd-iters", type=int, default=0, help="run this many Lloyd iterations first "
                    "(later iterations have better separated clusters than a random start)")
    ap.add_argument("--no-fp32", action="store_true")
    ap.add_argument("--eps", default="", help="comma list of TPQ_LL_EPS what-if bounds: share of listed points")
    ap.add_argument("--exp", default="", help="comma list of TPQ_LL_EXP kernel variants to time (assign only)")
    args = ap.parse_args()
    from torchpq_amd import kernels as K
    dev = "cuda:0"
    g = torch.Generator(device=dev)
    g.manual_seed(1237)
    l, d, n, k = args.l, args.d, args.n, args.k
    if args.data == "gauss":
        data = torch.randn(l, d, n, generator=g, device=dev)
    elif args.data == "sift":
        cen = torch.randn(l, d, 64, generator=g, device=dev).abs() * 40
        a = torch.randint(0, 64, (n,), generator=g, device=dev)
        data = (cen[:, :, a] + torch.randn(l, d, n, generator=g, device=dev) * 25).abs().round().clamp_(0, 218)
    else:
        raise SystemExit("unknown --data")
    cent = data[:, :, torch.randperm(n, generator=g, device=dev)[:k]].contiguous()

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

    out = {"config": vars(args)}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    step = K.LloydStepHip(data, cent)
    e1.record()
    torch.cuda.synchronize()
    out["prepare_ms_first"] = round(e0.elapsed_time(e1), 3)
    out["prepare_ms"] = round(timeit(lambda: K.LloydStepHip(data, cent), 2), 3)
    upd = K.ComputeCentroidsHip()
    for _ in range(args.lloyd_iters):
        _, _, cent = step(cent)
    vals, lab, new = step(cent)
    torch.cuda.synchronize()
    out["rechecked_share"] = round(float(step.rechecked().double().sum().item()) / (l * n), 6)
    out["level2_share"] = round(float(step.rechecked(1).double().sum().item()) / (l * n), 6)
    ref_new = upd(data, lab, k=k)
    scale = float(ref_new.abs().max().item())
    out["new_centroids_max_abs_diff_over_scale"] = float((new - ref_new).abs().max().item()) / scale
    if not args.no_fp32:
        v32, l32 = K.MaxSimHip(distance="euclidean")(data, cent, dim=2, mode="tn")
        out["labels_equal_fp32"] = float((lab == l32).double().mean().item())
        out["vals_max_rel_err"] = float(((vals - v32).abs().max() / v32.abs().max()).item())
        out["fp32_assign_ms"] = round(timeit(lambda: K.MaxSimHip(distance="euclidean")(data, cent, dim=2, mode="tn"), 2), 3)
    sel = K.MaxSimSelectHip(distance="euclidean")
    vs, ls = sel(data, cent)
    out["labels_equal_select"] = float((lab == ls).double().mean().item())
    out["old_select_ms"] = round(timeit(lambda: sel(data, cent)), 3)
    out["old_update_ms"] = round(timeit(lambda: upd(data, lab, k=k)), 3)
    out["step_ms"] = round(timeit(lambda: step(cent)), 3)
    out["step_assign_only_ms"] = round(timeit(lambda: step(cent, update=False)), 3)
    for eps in [x for x in args.eps.split(",") if x]:
        os.environ["TPQ_LL_EPS"] = eps
        step(cent, update=False)
        out[f"eps{eps}_rechecked_share"] = round(float(step.rechecked().double().sum().item()) / (l * n), 6)
    os.environ.pop("TPQ_LL_EPS", None)
    for e in [x for x in args.exp.split(",") if x]:
        os.environ["TPQ_LL_EXP"] = e
        v2, l2, _ = step(cent, update=False)
        out[f"exp{e}_labels_equal"] = float((l2 == lab).double().mean().item())
        out[f"exp{e}_rechecked_share"] = round(float(step.rechecked().double().sum().item()) / (l * n), 6)
        out[f"exp{e}_assign_only_ms"] = round(timeit(lambda: step(cent, update=False)), 3)
    os.environ.pop("TPQ_LL_EXP", None)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
