#!/usr/bin/env python
"""Per-stage timing of IVFPQIndex.search on a synthetic, directly injected index (random coarse
centroids / PQ codebooks / codes; no train or add), for shapes too big or too slow to train here.

    python tools/search_breakdown.py --preset c2|c3|c4 [--nq N] [--k K] [--residual]

Stages: probe (coarse GEMM + fused epilogue/select), tables (ADC LUT or residual part1; zero for the
fused-LUT path), scan (list scan + top-k + id map), total (one search() call, events on the stream).
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

PRESETS = {
    "c1": dict(d=128, m=16, n_cells=256, cell=391, nq=1000, n_probe=16, k=10),
    "c2": dict(d=128, m=64, n_cells=1024, cell=977, nq=10000, n_probe=32, k=100),
    "c3": dict(d=960, m=120, n_cells=1024, cell=977, nq=1000, n_probe=64, k=100),
    "c4": dict(d=128, m=64, n_cells=16384, cell=6103, nq=10000, n_probe=64, k=100),
}


def inject(idx, n_cells, cell, m, d, gen, slack=47):
    dev = idx.device
    sizes = torch.full((n_cells,), cell, device=dev, dtype=torch.long)
    cap = sizes + slack
    start = torch.cumsum(cap, 0) - cap
    n_slots = int(cap.sum().item())
    idx._storage = torch.randint(0, 256, (m // 4, n_slots, 4), generator=gen, device=dev,
                                 dtype=torch.uint8)
    idx._cell_start, idx._cell_size, idx._cell_capacity = start, sizes, cap
    pos = torch.arange(n_slots, device=dev)
    cell_of = torch.repeat_interleave(torch.arange(n_cells, device=dev), cap)
    live = pos < (start + sizes)[cell_of]
    idx._is_empty = (~live).to(torch.uint8)
    idx._address2id = torch.where(live, pos, torch.full_like(pos, -1))
    idx._max_id = n_slots
    idx._packed, idx._packed_valid, idx._has_holes = None, False, False
    idx.vq_codec.kmeans.register_buffer(
        "centroids", torch.rand(d, n_cells, generator=gen, device=dev) * 100)
    idx.vq_codec._trained(True)
    idx.pq_codec.kmeans.register_buffer(
        "centroids", torch.randn(m, d // m, 256, generator=gen, device=dev) * 20)
    idx.pq_codec._trained(True)


def timed(fn, iters):
    for _ in range(2):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        out = fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="c2", choices=sorted(PRESETS))
    ap.add_argument("--nq", type=int)
    ap.add_argument("--k", type=int)
    ap.add_argument("--n-probe", type=int)
    ap.add_argument("--m", type=int)
    ap.add_argument("--d", type=int)
    ap.add_argument("--n-cells", type=int)
    ap.add_argument("--cell", type=int, help="vectors per cell")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--residual", action="store_true")
    ap.add_argument("--no-fused", action="store_true")
    ap.add_argument("--ref-layout", action="store_true")
    ap.add_argument("--graph", action="store_true", help="also time a HIP-graph replay of search()")
    args = ap.parse_args()
    p = dict(PRESETS[args.preset])
    for key in ("nq", "k", "n_probe", "m", "d", "n_cells", "cell"):
        if getattr(args, key) is not None:
            p[key] = getattr(args, key)
    from torchpq_amd.index import IVFPQIndex
    dev = "cuda:0"
    gen = torch.Generator(device=dev)
    gen.manual_seed(0)
    idx = IVFPQIndex(d_vector=p["d"], n_subvectors=p["m"], n_cells=p["n_cells"], initial_size=8,
                     device=dev, pq_use_residual=args.residual)
    inject(idx, p["n_cells"], p["cell"], p["m"], p["d"], gen)
    idx.n_probe = p["n_probe"]
    idx.use_smart_probing = False
    idx.use_fused_lut = not args.no_fused
    idx.use_packed_layout = not args.ref_layout
    x = torch.rand(p["d"], p["nq"], generator=gen, device=dev) * 100
    k = p["k"]

    t_probe, (sims, cells, npl) = timed(lambda: idx.probe(x), args.iters)
    fused = idx.use_fused_lut and idx.d_subvector <= 4
    if fused:
        t_tab = 0.0  # the scan workgroups build their query's table themselves
    elif args.residual:
        t_tab, _ = timed(lambda: idx.precomputed_adc_residual_precomputed(x), args.iters)
    else:
        t_tab, _ = timed(lambda: idx.pq_codec.precompute_adc(x), args.iters)
    t_cells, _ = timed(lambda: idx.search_cells(x, cells, base_sims=sims, n_probe_list=npl, k=k),
                       args.iters)
    t_total, _ = timed(lambda: idx.search(x, k=k), args.iters)
    algo = int(idx._cell_size[cells].sum().item()) * p["m"]
    out = {"config": p, "residual": args.residual, "fused_lut": bool(idx.use_fused_lut and idx.d_subvector <= 4),
           "probe_ms": round(t_probe, 4), "tables_ms": round(t_tab, 4),
           "scan_ms": round(t_cells - t_tab, 4), "total_ms": round(t_total, 4),
           "qps": round(p["nq"] / t_total * 1e3, 1),
           "scan_GBps": round(algo / max(t_cells - t_tab, 1e-9) / 1e6, 1),
           "end_to_end_GBps": round(algo / t_total / 1e6, 1)}
    if args.graph:
        import time
        g = idx.graphed_search(p["nq"], k=k)
        t_graph, _ = timed(lambda: g(x), args.iters)
        out["graph_total_ms"] = round(t_graph, 4)
        out["graph_qps"] = round(p["nq"] / t_graph * 1e3, 1)
        # host-visible latency of one call (submit + wait), eager vs graph
        for name, fn in (("eager", lambda: idx.search(x, k=k)), ("graph", lambda: g(x))):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                fn()
                torch.cuda.synchronize()
            out[f"{name}_sync_latency_ms"] = round((time.perf_counter() - t0) / args.iters * 1e3, 4)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
