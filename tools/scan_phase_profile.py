#!/usr/bin/env python
"""Phase breakdown of scan_packed_kernel from in-kernel timestamps (private instrumentation build).

    EXTRA_FLAGS=-DTPQ_SCAN_PROFILE FORCE=1 bash torchpq_amd/csrc/build.sh
    python tools/scan_phase_profile.py [--m 64] [--nq 10000] [--n-probe 32] [--fused]

Prints the mean duration of each phase per workgroup (thread 0's view; 10 ns clock) and the share of
the kernel it accounts for.  Rebuild without the flag afterwards.
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PHASES = ["start->probe table", "LUT staged (+barrier)", "error bound", "scan loop", "final flush+publish",
          "barrier (wait slowest wave)", "counting rounds", "exact refinement + sort", "store lists"]
FUSED_PHASES = PHASES[:8] + ["workgroup tree merge", "store + release + ticket", "(last) fold the splits + tree",
                             "(last) write + overflow test"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--n-cells", type=int, default=1024)
    ap.add_argument("--cell", type=int, default=977)
    ap.add_argument("--n-probe", type=int, default=32)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--fused", action="store_true")
    ap.add_argument("--ds", type=int, default=2, help="sub-vector length of the fused table")
    ap.add_argument("--n-split", type=int, help="workgroups per query (default: the wrapper's heuristic)")
    args = ap.parse_args()
    from torchpq_amd import kernels as K
    from torchpq_amd import _lib
    lib = _lib.load()
    raw = ctypes.CDLL(lib._name)
    dev = "cuda:0"
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    m, nc, nq = args.m, args.n_cells, args.nq
    sizes = torch.full((nc,), args.cell, device=dev, dtype=torch.long)
    cap = sizes + 47
    start = torch.cumsum(cap, 0) - cap
    n_slots = int(cap.sum().item())
    storage = torch.randint(0, 256, (m // 4, n_slots, 4), generator=g, device=dev, dtype=torch.uint8)
    cells = torch.rand(nq, nc, generator=g, device=dev).argsort(1)[:, :args.n_probe].contiguous()
    cs, sz = start[cells].contiguous(), sizes[cells].contiguous()
    npl = torch.full((nq,), args.n_probe, device=dev, dtype=torch.long)
    scan = K.IVFPQTopkHip(m=m)
    packed = K.PackCodesHip()(storage)
    if args.fused:
        cb = torch.randn(m, args.ds, 256, generator=g, device=dev) * 20
        q = torch.randn(args.ds * m, nq, generator=g, device=dev) * 20
        run = lambda: scan.topk_fused(storage, q, cb, None, cs, sz, npl, n_candidates=args.k, packed=packed,
                                      n_split=args.n_split, slots_hint=args.n_probe * args.cell)
    else:
        lut = torch.randn(m, nq, 256, generator=g, device=dev) * 50 - 300
        run = lambda: scan.topk(storage, lut, None, cs, sz, npl, n_candidates=args.k, packed=packed,
                                n_split=args.n_split, slots_hint=args.n_probe * args.cell)
    for _ in range(2):
        run()
    prof = torch.zeros(nq * 64 * 16, device=dev, dtype=torch.int64)
    raw.tpq_debug_set_scan_profile(ctypes.c_void_p(prof.data_ptr()))
    run()
    torch.cuda.synchronize()
    raw.tpq_debug_set_scan_profile(None)
    n_blocks = nq * scan.last_n_split
    raw_t = prof.view(-1, 16)[:n_blocks].double().cpu() * 10.0  # ns
    # (the fused finish stamps slots 7-12; 11, 12 in the finishing workgroup only; the dump modes end at slot 6 and use
    # slots 10 ... 14 for the sub-phases of their prologue)
    dump = bool((raw_t[:, 6] > 0).any()) and not bool((raw_t[:, 7] > 0).any())
    fused = bool((raw_t[:, 12] > 0).any()) and not dump
    if fused:
        fin = raw_t[:, 12] > 0
        t = raw_t[:, :11]
        d = (t[:, 1:] - t[:, :-1]) / 1e3
        total = (t[:, 10] - t[:, 0]) / 1e3
        print(f"m={m} nq={nq} n_probe={args.n_probe} cell={args.cell} k={args.k} blocks={n_blocks} (fused finish)  mean block lifetime to the ticket {total.mean():.1f} us")
        for i, name in enumerate(FUSED_PHASES[:10]):
            print(f"  {name:32s} {d[:, i].mean():7.2f} us")
        tf = raw_t[fin]
        print(f"  {FUSED_PHASES[10]:32s} {((tf[:, 11] - tf[:, 10]) / 1e3).mean():7.2f} us  (finishing workgroups: {int(fin.sum())})")
        print(f"  {FUSED_PHASES[11]:32s} {((tf[:, 12] - tf[:, 11]) / 1e3).mean():7.2f} us")
        print(f"  kernel span {(raw_t[:, :13].max() - raw_t[:, 0].min()) / 1e3:.1f} us")
        return
    if dump:
        # dump mode (large batches, m = 64): the workgroup ends after the loop; scan_finish_exact_kernel does the rest
        t = raw_t[:, :7]
        d = (t[:, 1:] - t[:, :-1]) / 1e3
        total = (t[:, 6] - t[:, 0]) / 1e3
        slots = 1024
        print(f"m={m} nq={nq} n_probe={args.n_probe} cell={args.cell} k={args.k} blocks={n_blocks} (dump mode)  mean block lifetime {total.mean():.1f} us")
        for i, name in enumerate(["start->probe loads issued", "table computed, quantised, staged (+barriers)", "error bound",
                                  "scan loop", "final flush", "store the list"]):
            print(f"  {name:44s} {d[:, i].mean():7.2f} us  {100 * d[:, i].mean() / total.mean():5.1f} %")
        span = (t[:, 6].max() - t[:, 0].min()) / 1e3
        print(f"  kernel span {span:.1f} us; sum of block lifetimes / ({slots} slots) = {total.sum() / slots:.1f} us")
        if bool((raw_t[:, 13] > 0).any()):  # sub-phases of the prologue (slots 10 ... 14), thread 0's view
            us = lambda a, b: float(((raw_t[:, b] - raw_t[:, a]) / 1e3).mean())  # noqa: E731
            print("  prologue in detail:")
            sel16 = bool((raw_t[:, 12] > 0).any())
            rows = [("query staged (global gather + barrier)", 1, 10),
                    ("table entries computed (codebook loads, fma, maxima)", 10, 11)]
            rows += [("probe table + barrier (the slowest wave's entries)", 11, 12),
                     ("maxima reduced, entries quantised and stored", 12, 13)] if sel16 else \
                    [("probe table (wave 0)", 11, 13)]
            rows += [("barrier after the stores", 13, 2), ("error bound, list init", 2, 3)]
            if bool((raw_t[:, 14] > 0).any()):
                rows.append(("first tile: located, loaded, consumed", 3, 14))
            for name, a, b in rows:
                print(f"    {name:56s} {us(a, b):7.2f} us")
        return
    t = raw_t[:, :10]
    d = (t[:, 1:] - t[:, :-1]) / 1e3  # us
    total = (t[:, 9] - t[:, 0]) / 1e3
    print(f"m={m} nq={nq} n_probe={args.n_probe} blocks={n_blocks}  mean block lifetime {total.mean():.1f} us")
    for i, name in enumerate(PHASES):
        print(f"  {name:32s} {d[:, i].mean():7.2f} us  {100 * d[:, i].mean() / total.mean():5.1f} %")
    span = (t[:, 9].max() - t[:, 0].min()) / 1e3
    print(f"  kernel span {span:.1f} us; sum of block lifetimes / (512 slots) = {total.sum() / 512:.1f} us")


if __name__ == "__main__":
    main()
