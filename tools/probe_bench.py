#!/usr/bin/env python
"""The coarse step of search() (tpq_ivfpq_coarse_probe_route) on its three routes: time per call and equality of the
results.   python tools/probe_bench.py [--nq 10000] [--d 128] [--n-cells 4096,16384] [--n-probe 1,16,128]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--n-cells", default="1024,4096,16384")
    ap.add_argument("--n-probe", default="1,16,128")
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    import bench
    from torchpq_amd import kernels as K
    dev = torch.device("cuda", 0)
    synth = bench.SiftLike(args.d, dev)
    x = synth.sample(args.nq, seed=4321)
    out = []
    for n_cells in [int(v) for v in args.n_cells.split(",")]:
        c = synth.sample(n_cells, seed=7)
        z = torch.zeros(n_cells, device=dev, dtype=torch.long)
        for n_probe in [int(v) for v in args.n_probe.split(",")]:
            rec = {"nq": args.nq, "d": args.d, "n_cells": n_cells, "n_probe": n_probe}
            ref = None
            prepared = K.CoarseProbeHip.prepare(c)   # (once per codebook, as IVFPQIndex does)
            for route in ("fp32", "fp16", "auto", "fp16_unprepared"):
                op = K.CoarseProbeHip(route=route.split("_")[0])
                prep = None if route.endswith("unprepared") else prepared
                for _ in range(2):
                    r = op(x, c, z, z, n_probe, None, prepared=prep)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.iters):
                    r = op(x, c, z, z, n_probe, None, prepared=prep)
                e1.record()
                torch.cuda.synchronize()
                rec[f"{route}_ms"] = round(e0.elapsed_time(e1) / args.iters, 4)
                if ref is None:
                    ref = r
                else:
                    rec[f"{route}_equal_fp32"] = bool(torch.equal(ref[0], r[0]) and torch.equal(ref[1], r[1]))
            out.append(rec)
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
