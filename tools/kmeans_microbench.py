#!/usr/bin/env python
"""MultiKMeans kernels at BASELINE.json configs[4]: n_kmeans=64 d=64 n=1M k=256 (PQ codebook learn).
Reports the assign kernel (tpq_max_sim, fp32 MFMA) in TFLOP/s and the update kernel in GB/s."""
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
    ap.add_argument("--rows-read", action="store_true", help="also time the update's bare read pattern")
    args = ap.parse_args()
    from torchpq_amd import kernels as K
    dev = "cuda:0"
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    data = torch.randn(args.l, args.d, args.n, generator=g, device=dev)
    cent = data[:, :, torch.randperm(args.n, generator=g, device=dev)[:args.k]].contiguous()
    ms_k, cc_k = K.MaxSimHip(distance="euclidean"), K.ComputeCentroidsHip()
    v, lab = ms_k(data, cent, dim=2, mode="tn")
    cc_k(data, lab, k=args.k)
    torch.cuda.synchronize()

    def timeit(fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / args.iters

    t_assign = timeit(lambda: ms_k(data, cent, dim=2, mode="tn"))
    t_update = timeit(lambda: cc_k(data, lab, k=args.k))
    flop = 2.0 * args.l * args.n * args.k * args.d
    byt = 4.0 * args.l * args.d * args.n
    extra = {}
    if args.rows_read:
        from torchpq_amd import _lib
        lib = _lib.load()
        for chunks in (8, 32, 128):
            t = timeit(lambda: _lib.check(lib.tpq_ubench_rows_read(
                _lib.ptr(data), args.l, args.d, args.n, chunks, None, _lib.stream_ptr(dev)), "rows_read"))
            extra[f"rows_read_chunks{chunks}_GBps"] = round(byt / t / 1e6, 1)
    print(json.dumps({
        "config": vars(args), **extra,
        "assign_ms": round(t_assign, 3), "assign_TFLOPs": round(flop / t_assign / 1e9, 2),
        "assign_GBps": round(byt / t_assign / 1e6, 1),
        "update_ms": round(t_update, 3),
        "update_GBps": round((byt + 8.0 * args.l * args.n) / t_update / 1e6, 1),
        "iter_ms": round(t_assign + t_update, 3)}))


if __name__ == "__main__":
    main()
