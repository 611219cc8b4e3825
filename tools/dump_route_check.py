#!/usr/bin/env python
"""The large-batch routes of the packed scan (dump mode + scan_finish_exact_kernel, csrc/scan_device.h) against the
reference-layout kernel -- values and addresses bit for bit -- and their time, on synthetic uniform indexes; both
LUT sources (the materialised table, the table built in the workgroup from query + codebook).

    python tools/dump_route_check.py [--quick]
With a variant library (TPQ_AMD_LIB=torchpq_amd/variants/libtorchpq_amd_ab.so) TPQ_SCAN_DUMP=0/1/2 forces the route.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(K, m, ds, nc, cell, n_probe, k, nq, fused, iters, check, skew=False, holes=False):
    dev = "cuda:0"
    g = torch.Generator(device=dev)
    g.manual_seed(m * 1000 + n_probe + k)
    if skew:   # cell sizes from 0 to 4 x the mean
        sizes = (torch.rand(nc, generator=g, device=dev) ** 2 * 3 * cell).long()
    else:
        sizes = torch.full((nc,), cell, device=dev, dtype=torch.long)
    cap = sizes + 47
    start = torch.cumsum(cap, 0) - cap
    n_slots = int(cap.sum().item())
    storage = torch.randint(0, 256, (m // 4, n_slots, 4), generator=g, device=dev, dtype=torch.uint8)
    is_empty = None
    if holes:
        is_empty = (torch.rand(n_slots, generator=g, device=dev) < 0.1).to(torch.uint8)
    codebook = torch.randn(m, ds, 256, generator=g, device=dev) * 30 + 60
    query = torch.randn(m * ds, nq, generator=g, device=dev) * 30 + 60
    cells = torch.rand(nq, nc, generator=g, device=dev).argsort(1)[:, :n_probe].contiguous()
    cs, sz = start[cells].contiguous(), sizes[cells].contiguous()
    npl = torch.full((nq,), n_probe, device=dev, dtype=torch.long)
    scan = K.IVFPQTopkHip(m=m)
    packed = K.PackCodesHip()(storage)
    lut = None if fused else K.AdcLutHip()(query, codebook)
    hint = n_probe * cell

    def call(pk):
        if fused:
            return scan.topk_fused(storage, query, codebook, is_empty, cs, sz, npl, k, packed=pk, slots_hint=hint)
        return scan.topk(storage, lut, is_empty, cs, sz, npl, n_candidates=k, packed=pk, slots_hint=hint)

    out = {"m": m, "ds": ds, "cells": nc, "cell": cell, "n_probe": n_probe, "k": k, "nq": nq, "fused": fused,
           "skew": skew, "holes": holes}
    r = call(packed)
    torch.cuda.synchronize()
    if check:
        e = call(None)
        torch.cuda.synchronize()
        out["equal"] = bool(torch.equal(r[0], e[0]) and torch.equal(r[1], e[1]))
        if not out["equal"]:
            bad = (r[1] != e[1]).any(1) | (r[0] != e[0]).any(1)
            out["bad_queries"] = int(bad.sum().item())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call(packed)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    out["ms"] = round(ms, 4)
    out["GBps"] = round(float(sz.sum().item()) * m / ms / 1e6, 1)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--large-k", action="store_true", help="k in (248, 504] at the C2 shape (and two shapes the route leaves to the lists)")
    ap.add_argument("--large-k-sweep", action="store_true")
    ap.add_argument("--one", default=None, help="m,ds,cells,cell,n_probe,k,nq: that shape only (fused, no check)")
    args = ap.parse_args()
    from torchpq_amd import kernels as K
    if args.one:
        print(json.dumps(run(K, *[int(x) for x in args.one.split(",")], True, args.iters, False)), flush=True)
        return
    if args.large_k_sweep:  # where lists of 4 registers in 4-wave workgroups stop paying: slots per query x k
        for k in (300, 500):
            for cell, n_probe in ((244, 16), (244, 32), (244, 64), (977, 8), (977, 16), (977, 24), (977, 32), (977, 64)):
                print(json.dumps(run(K, 64, 2, 4096 if cell == 244 else 1024, cell, n_probe, k, 10000, True, args.iters,
                                     False)), flush=True)
        return
    if args.large_k:
        for k in (200, 249, 300, 400, 500, 504):
            print(json.dumps(run(K, 64, 2, 1024, 977, 32, k, 10000, True, args.iters, not args.no_check)), flush=True)
        print(json.dumps(run(K, 64, 2, 4096, 244, 32, 300, 10000, True, args.iters, not args.no_check)), flush=True)
        print(json.dumps(run(K, 64, 1, 1024, 600, 20, 450, 3000, True, 5, not args.no_check, skew=True, holes=True)), flush=True)
        return
    shapes = [  # m, ds, cells, cell, n_probe, k, nq
        (64, 2, 1024, 977, 32, 100, 10000), (64, 2, 4096, 244, 16, 100, 10000), (64, 2, 4096, 244, 8, 100, 10000),
        (64, 2, 16384, 61, 32, 100, 10000), (64, 2, 4096, 244, 64, 100, 10000), (64, 2, 16384, 61, 128, 100, 10000),
        (64, 2, 4096, 244, 1, 100, 10000), (64, 2, 4096, 244, 16, 1, 10000), (64, 2, 4096, 244, 16, 10, 10000),
        (64, 2, 4096, 244, 32, 200, 10000), (64, 4, 1024, 977, 16, 100, 3000), (64, 1, 1024, 977, 16, 248, 1500),
        (64, 2, 16384, 6103, 64, 100, 2000),
    ]
    if args.quick:
        shapes = shapes[:4]
    for sh in shapes:
        for fused in (True, False):
            print(json.dumps(run(K, *sh, fused, args.iters, not args.no_check)), flush=True)
    if not args.quick and not args.no_check:
        print(json.dumps(run(K, 64, 2, 1024, 500, 24, 100, 4000, True, 5, True, skew=True, holes=True)), flush=True)
        print(json.dumps(run(K, 64, 2, 2048, 30, 40, 100, 2000, False, 5, True, skew=True, holes=True)), flush=True)


if __name__ == "__main__":
    main()
