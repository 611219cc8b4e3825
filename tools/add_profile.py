#!/usr/bin/env python
"""Wall time of IVFPQIndex.train / add at the SIFT1M shape, stage by stage (torch events).

    python tools/add_profile.py [--n 1000000] [--m 64] [--n-cells 1024]
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1000000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--n-cells", type=int, default=1024)
    ap.add_argument("--batches", type=int, default=2)
    args = ap.parse_args()
    from torchpq_amd.index import IVFPQIndex
    dev = "cuda:0"
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    centers = torch.rand(args.d, 256, generator=g, device=dev) * 100
    def data(n):
        a = torch.randint(0, 256, (n,), generator=g, device=dev)
        return (centers[:, a] + torch.randn(args.d, n, generator=g, device=dev) * 12).abs().round().contiguous()
    idx = IVFPQIndex(d_vector=args.d, n_subvectors=args.m, n_cells=args.n_cells, initial_size=2048, device=dev)
    t_train, _ = timed(lambda: idx.train(data(100000)))
    out = {"train_ms": round(t_train, 1), "adds": []}
    next_id = 0
    for b in range(args.batches):
        x = data(args.n)
        ids = torch.arange(next_id, next_id + args.n, device=dev)
        next_id += args.n
        t_vq, cells = timed(lambda: idx.vq_codec.encode(x))
        t_pq, codes = timed(lambda: idx.pq_codec.encode(x))
        t_cont, _ = timed(lambda: super(IVFPQIndex, idx).add(codes, cells, ids=ids))
        x2 = data(args.n)
        ids2 = torch.arange(next_id, next_id + args.n, device=dev)
        next_id += args.n
        t_add, _ = timed(lambda: idx.add(x2, ids=ids2))
        out["adds"].append({"coarse_assign_ms": round(t_vq, 2), "pq_encode_ms": round(t_pq, 2),
                            "container_add_ms": round(t_cont, 2), "index_add_ms": round(t_add, 2),
                            "n_items": int(idx.n_items), "capacity": int(idx.capacity)})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
