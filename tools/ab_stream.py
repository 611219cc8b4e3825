#!/usr/bin/env python
"""Same-box A/B of scan variants on the streaming regimes (round 6): the C2 shape (cache-fed), the 100 M-slot index
with the timed batch (39 reads of each byte per launch) and COLD (every cell probed exactly once per launch: DRAM).
    TPQ_AMD_LIB=torchpq_amd/variants/libtorchpq_amd_<name>.so python tools/ab_stream.py
prints one JSON line per regime: scan-call ms (HIP events around the scan), TB/s of algorithmic bytes."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def timed(idx, fn, steps):
    scan = idx._ivfpq_topk._scan
    fn()
    torch.cuda.synchronize()
    scan.record_events = []
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    per = [a.elapsed_time(b) for a, b in scan.record_events]
    scan.record_events = None
    return float(np.median(per)), float(min(per))


def main():
    dev = torch.device("cuda:0")
    out = {"lib": os.environ.get("TPQ_AMD_LIB", "product")}
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    # C2 shape, fabricated (uniform random codes): 10 000 x 32 x ~977
    idx = bench.fabricate_index(dev, 128, 64, 1024, 1_000_000, seed=11)
    idx.n_probe, idx.use_smart_probing = 32, False
    q = torch.randn(128, 10000, generator=g, device=dev)
    med, mn = timed(idx, lambda: idx.search(q, k=100), 20)
    algo = bench.scanned_bytes(idx, q, 64)
    out["c2"] = {"ms": round(med, 4), "min_ms": round(mn, 4), "TBps": round(algo / med / 1e9, 3)}
    del idx
    torch.cuda.empty_cache()
    idx = bench.fabricate_index(dev, 128, 64, 16384, 100_000_000, seed=1236)
    idx.n_probe, idx.use_smart_probing = 64, False
    med, mn = timed(idx, lambda: idx.search(q, k=100), 5)
    algo = bench.scanned_bytes(idx, q, 64)
    out["c4_warm"] = {"ms": round(med, 4), "min_ms": round(mn, 4), "TBps": round(algo / med / 1e9, 3)}
    total = int(idx._cell_size.sum().item()) * 64
    for nq in (1024, 2048, 4096, 8192):
        n_probe = 16384 // nq
        cells = torch.randperm(16384, generator=g, device=dev).view(nq, n_probe).contiguous()
        qq = torch.randn(128, nq, generator=g, device=dev)
        npl = torch.full((nq,), n_probe, device=dev, dtype=torch.long)
        med, mn = timed(idx, lambda: idx.search_cells(qq, cells, n_probe_list=npl, k=100), 10)
        out[f"c4_cold_{nq}x{n_probe}"] = {"ms": round(med, 4), "min_ms": round(mn, 4), "TBps": round(total / med / 1e9, 3),
                                          "route": idx._ivfpq_topk._scan.last_route()}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
