#!/usr/bin/env python
"""tpq_max_sim_split (bf16 x 3 split on the bf16 matrix cores) against float64 and against the
bit-exact fp32 kernel: error of the returned maxima relative to the scale sum|a_k c_k| + |a|^2 + |c|^2,
label agreement, every disagreement checked to be a near-tie in float64, and the time of both at
the C5 shape (BASELINE.json configs[4])."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def check_case(K, l, d, m, n, distance, scale=1.0, seed=0):
    dev = "cuda:0"
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    data = torch.randn(l, d, m, generator=g, device=dev) * scale
    cent = data[:, :, torch.randperm(m, generator=g, device=dev)[:n] % m].contiguous()
    cent = cent + 0.1 * scale * torch.randn(cent.shape, generator=g, device=dev)
    exact = K.MaxSimHip(distance=distance)
    split = K.MaxSimHip(distance=distance, precision="bf16x3")
    assert split.split_supported(d, m, n)
    ve, ie = exact(data, cent, dim=2)
    vs, is_ = split(data, cent, dim=2)
    a64, c64 = data.double(), cent.double()
    dots = torch.einsum("ldm,ldn->lmn", a64, c64)
    if distance == "euclidean":
        sims = 2 * dots - (a64 * a64).sum(1)[:, :, None] - (c64 * c64).sum(1)[:, None, :]
        mag = 2 * torch.einsum("ldm,ldn->lmn", a64.abs(), c64.abs()) + (a64 * a64).sum(1)[:, :, None] \
            + (c64 * c64).sum(1)[:, None, :]
    else:
        sims = dots
        mag = torch.einsum("ldm,ldn->lmn", a64.abs(), c64.abs())
    v64, i64 = sims.max(dim=2)
    out = {}
    for name, v, i in (("fp32", ve, ie), ("bf16x3", vs, is_)):
        scale_at = mag.gather(2, i[:, :, None])[:, :, 0]
        err = ((v.double() - sims.gather(2, i[:, :, None])[:, :, 0]).abs() / scale_at).max().item()
        mism = i != i64
        gap = ((v64 - sims.gather(2, i[:, :, None])[:, :, 0]) / scale_at)[mism]
        out[name] = {"max_rel_err": err, "label_mismatch_vs_f64": int(mism.sum().item()),
                     "worst_mismatch_gap": float(gap.max().item()) if gap.numel() else 0.0}
    out["labels_equal_fp32_vs_split"] = float((ie == is_).double().mean().item())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--no-time", action="store_true")
    args = ap.parse_args()
    from torchpq_amd import kernels as K
    res = {}
    for (l, d, m, n, dist, scale) in [(3, 64, 5000, 256, "euclidean", 1.0), (2, 64, 777, 200, "euclidean", 100.0),
                                      (2, 17, 3000, 256, "euclidean", 1.0), (1, 48, 2049, 300, "euclidean", 1e-3),
                                      (2, 33, 1000, 64, "inner", 1.0), (1, 8, 70000, 256, "euclidean", 1.0)]:
        res[f"l{l}_d{d}_m{m}_n{n}_{dist}_s{scale}"] = check_case(K, l, d, m, n, dist, scale)
    if not args.no_time:
        dev = "cuda:0"
        g = torch.Generator(device=dev)
        g.manual_seed(0)
        L, D, N, KK = 64, 64, 1000000, 256
        data = torch.randn(L, D, N, generator=g, device=dev)
        cent = data[:, :, :KK].contiguous()
        for prec in ("fp32", "bf16x3"):
            k = K.MaxSimHip(distance="euclidean", precision=prec)
            k(data, cent, dim=2)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.iters):
                k(data, cent, dim=2)
            e1.record()
            torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / args.iters
            res[f"c5_{prec}"] = {"ms": round(t, 3), "TFLOPs_fp32_equivalent": round(2.0 * L * N * KK * D / t / 1e9, 1)}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
