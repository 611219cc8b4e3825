#!/usr/bin/env python
"""Tune bench.py's synthetic SIFT stand-in: recall_gt@100 of the C2 configuration (n_cells=1024,
m=64, n_probe=32) must be diagnostic (0.90-0.99, the reference's SIFT1M figure is 0.95), not 1.0.
Prints one JSON line per generator setting."""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="default")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    grid = [
        dict(n_centers=256, latent_dim=12, latent_scale=55.0, noise=12.0),
        dict(n_centers=256, latent_dim=12, latent_scale=55.0, noise=20.0),
        dict(n_centers=256, latent_dim=24, latent_scale=55.0, noise=12.0),
        dict(n_centers=256, latent_dim=24, latent_scale=80.0, noise=12.0),
        dict(n_centers=256, latent_dim=48, latent_scale=80.0, noise=12.0),
        dict(n_centers=2048, latent_dim=12, latent_scale=55.0, noise=12.0),
        dict(n_centers=2048, latent_dim=24, latent_scale=80.0, noise=12.0),
        dict(n_centers=2048, latent_dim=48, latent_scale=80.0, noise=20.0),
        dict(n_centers=16, latent_dim=32, latent_scale=100.0, noise=10.0),
        dict(n_centers=16, latent_dim=64, latent_scale=100.0, noise=10.0),
        dict(n_centers=256, latent_dim=1, latent_scale=0.0, noise=30.0),   # round-1 generator
        dict(n_centers=256, latent_dim=1, latent_scale=0.0, noise=45.0),
    ]
    if a.grid != "default":
        grid = json.loads(a.grid)
    args = bench.parse_args(["--no-secondary"])
    for g in grid:
        t0 = time.time()
        synth = bench.SiftLike(args.d, dev, **g)
        base = synth.sample(args.n_base, seed=1)
        gs = torch.Generator(device=dev)
        gs.manual_seed(2)
        train = base[:, torch.randperm(args.n_base, generator=gs, device=dev)[:args.n_train]].contiguous()
        idx, t_train, t_add = bench.build_index(args, dev, base, train)
        idx.n_probe, idx.use_smart_probing = args.n_probe, False
        q = synth.sample(1000, seed=4321)
        nn = bench.exact_nn(q, base)
        out = dict(g)
        for npb in (8, 32, 64):
            idx.n_probe = npb
            v, ids = idx.search(q, k=100)
            out[f"r100@np{npb}"] = round(float((ids == nn[:, None]).any(1).float().mean()), 4)
            out[f"r1@np{npb}"] = round(float((ids[:, 0] == nn).float().mean()), 4)
        idx.n_probe = 32
        out["bytes_per_query"] = bench.scanned_bytes(idx, q, args.m) / 1000
        cs = idx._cell_size.double()
        out["cell_imbalance"] = round(float((cs ** 2).sum() * args.n_cells / cs.sum() ** 2), 3)
        out["frac_zero"] = round(float((base == 0).float().mean()), 3)
        out["frac_clamped"] = round(float((base == 218).float().mean()), 3)
        out["wall_s"] = round(time.time() - t0, 1)
        print(json.dumps(out), flush=True)
        del idx, base, train


if __name__ == "__main__":
    main()
