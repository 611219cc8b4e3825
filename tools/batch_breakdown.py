#!/usr/bin/env python
"""Where a small / middling batch spends its time (round 6): per batch size, wall time per search(),
the scan call's HIP-event time, the same search replayed from a HIP graph (GraphedSearch), at the C2 shape.
    python tools/batch_breakdown.py [--sizes 256,512,1250,2500,10000]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="256,512,1250,2500,5000,10000")
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    idx = bench.fabricate_index(dev, 128, 64, 1024, 1_000_000, seed=11)
    idx.n_probe, idx.use_smart_probing = 32, False
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    scan = idx._ivfpq_topk._scan
    full = None
    for nq in [int(x) for x in args.sizes.split(",")][::-1]:
        q = torch.randn(128, nq, generator=g, device=dev)
        for _ in range(3):
            idx.search(q, k=100)
        torch.cuda.synchronize()
        scan.record_events = []
        t0 = time.perf_counter()
        for _ in range(args.iters):
            idx.search(q, k=100)
        t_issue = time.perf_counter() - t0
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / args.iters
        scan_ms = float(np.median([a.elapsed_time(b) for a, b in scan.record_events]))
        scan.record_events = None
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            idx.search(q, k=100)
        e1.record()
        torch.cuda.synchronize()
        gpu_ms = e0.elapsed_time(e1) / args.iters
        gs = idx.graphed_search(nq, k=100)
        gs(q)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(args.iters):
            gs(q)
        e1.record()
        torch.cuda.synchronize()
        graph_ms = e0.elapsed_time(e1) / args.iters
        rec = {"nq": nq, "route": scan.last_route(), "n_split": scan.last_n_split,
               "wall_ms": round(wall * 1e3, 4), "host_issue_ms": round(t_issue / args.iters * 1e3, 4),
               "gpu_ms_back_to_back": round(gpu_ms, 4), "scan_call_ms": round(scan_ms, 4),
               "graph_replay_ms": round(graph_ms, 4), "Mqps_wall": round(nq / wall / 1e6, 3),
               "Mqps_graph": round(nq / graph_ms / 1e3, 3)}
        if full is None:
            full = rec
        rec["share_of_full_rate_wall"] = round(rec["Mqps_wall"] / full["Mqps_wall"], 3)
        rec["share_of_full_rate_graph"] = round(rec["Mqps_graph"] / full["Mqps_graph"], 3)
        print(json.dumps(rec), flush=True)
        del gs


if __name__ == "__main__":
    main()
