#!/usr/bin/env python
"""Microbenchmark of the list-scan kernels on a synthetic, directly generated index
(uniform cells, random codes, random LUT): isolates the kernel from train/add.

    python tools/scan_microbench.py [--nq 10000] [--n-cells 1024] [--cell 977] [--m 64]
                                    [--n-probe 32] [--k 100] [--layouts packed,ref] [--check]
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--n-cells", type=int, default=1024)
    ap.add_argument("--cell", type=int, default=977)
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--n-probe", type=int, default=32)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--layouts", default="packed,ref")
    ap.add_argument("--n-split", type=int, default=None)
    ap.add_argument("--neighbors", action="store_true",
                    help="queries probe a window of adjacent cells (correlated probe lists)")
    ap.add_argument("--sort", action="store_true", help="with --neighbors: sort queries by first cell")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--conflict-free", action="store_true",
                    help="m = 4 / 8 / 16: codes crafted so that the look-ups of 32 consecutive slots fall into 32 "
                         "distinct LDS banks (c mod (32 / m) = (slot / m) mod (32 / m)): what a replicated table would buy")
    args = ap.parse_args()
    from torchpq_amd import kernels as K
    dev = "cuda:0"
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    m, nc = args.m, args.n_cells
    sizes = torch.full((nc,), args.cell, device=dev, dtype=torch.long)
    cap = sizes + 47
    start = torch.cumsum(cap, 0) - cap
    n_slots = int(cap.sum().item())
    storage = torch.randint(0, 256, (m // 4, n_slots, 4), generator=g, device=dev, dtype=torch.uint8)
    if args.conflict_free:
        reps = 32 // m
        sl = torch.arange(n_slots, device=dev)
        low = ((sl // m) % reps).to(torch.uint8)[None, :, None]
        storage = (storage // reps) * reps + low
    lut = torch.randn(m, args.nq, 256, generator=g, device=dev) * 50 - 300
    if args.neighbors:
        base = torch.randint(0, nc, (args.nq, 1), generator=g, device=dev)
        if args.sort:
            base = base.sort(0).values
        cells = (base + torch.arange(args.n_probe, device=dev)[None, :]) % nc
    else:
        cells = torch.rand(args.nq, nc, generator=g, device=dev).argsort(1)[:, :args.n_probe].contiguous()
    cs, sz = start[cells].contiguous(), sizes[cells].contiguous()
    npl = torch.full((args.nq,), args.n_probe, device=dev, dtype=torch.long)
    scan = K.IVFPQTopkHip(m=m)
    packed = K.PackCodesHip()(storage)
    algo_bytes = int(sz.sum().item()) * m
    out = {}
    ref = None
    for layout in args.layouts.split(","):
        pk = packed if layout == "packed" else None
        for _ in range(2):
            r = scan.topk(storage, lut, None, cs, sz, npl, n_candidates=args.k, packed=pk,
                          n_split=args.n_split, slots_hint=args.n_probe * args.cell)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            r = scan.topk(storage, lut, None, cs, sz, npl, n_candidates=args.k, packed=pk,
                          n_split=args.n_split, slots_hint=args.n_probe * args.cell)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / args.iters
        out[layout] = {"ms": round(ms, 4), "GBps": round(algo_bytes / ms / 1e6, 1),
                       "Mqps": round(args.nq / ms / 1e3, 3)}
        if args.check:
            if ref is None:
                ref = r
            else:
                out[layout]["equal_to_first"] = bool(torch.equal(ref[0], r[0]) and torch.equal(ref[1], r[1]))
    out["bytes_per_query"] = algo_bytes / args.nq
    print(json.dumps(out))


if __name__ == "__main__":
    main()
