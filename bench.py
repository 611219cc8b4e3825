#!/usr/bin/env python
"""bench.py -- queries/sec + recall@100 of IVFPQIndex.search on a SIFT1M-shaped index.

Workload (BASELINE.json configs[1]; SIFT1M files are not available offline, so the data is
synthetic of the same shape -- SURVEY.md 8d "SIFT1M-like"): d=128 non-negative integer-valued
clustered fp32, 1 M base vectors, 100 k training vectors, 10 000 queries, IVFPQ n_cells=1024,
m=64 (8-bit), n_probe=32, k=100, use_smart_probing=False (deterministic scanned bytes).

A "step" is one search() call over the whole resident query batch (coarse GEMM + select + LUT +
list scan + id map).  `value` = queries/s over all ranks; each rank owns a replica of the index
(built on rank 0, broadcast once over RCCL) and its own 10 000 queries -> weak scaling, no
per-query collective.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--nq 10000] [--n-base 1000000]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable


def sift_like(gen, d, n, centers, device, noise=30.0):
    """clamp(round(|center + noise|)) in [0, 218]: non-negative, integer-valued, clustered."""
    out = torch.empty(d, n, device=device, dtype=torch.float32)
    step = 1 << 18
    for b in range(0, n, step):
        e = min(n, b + step)
        a = torch.randint(0, centers.shape[1], (e - b,), generator=gen, device=device)
        x = centers[:, a] + torch.randn(d, e - b, generator=gen, device=device) * noise
        out[:, b:e] = x.abs().round().clamp_(0, 218)
    return out


def make_centers(gen, d, device):
    # broad, overlapping mixture: k-means cells come out mildly unbalanced, as on real SIFT
    return torch.randn(d, 256, generator=gen, device=device).abs() * 40.0


def build_index(args, device):
    from torchpq_amd.index import IVFPQIndex
    gen = torch.Generator(device=device)
    gen.manual_seed(1234)
    centers = make_centers(gen, args.d, device)
    base = sift_like(gen, args.d, args.n_base, centers, device)
    np.random.seed(1234)
    idx = IVFPQIndex(d_vector=args.d, n_subvectors=args.m, n_cells=args.n_cells,
                     initial_size=max(64, 2 * args.n_base // args.n_cells), device=str(device))
    t0 = time.time()
    train = base[:, torch.randperm(args.n_base, generator=gen, device=device)[:args.n_train]].contiguous()
    idx.train(train)
    torch.cuda.synchronize()
    t_train = time.time() - t0
    t0 = time.time()
    for b in range(0, args.n_base, 1 << 18):
        idx.add(base[:, b:b + (1 << 18)].contiguous())
    torch.cuda.synchronize()
    t_add = time.time() - t0
    return idx, base, centers, gen, t_train, t_add


def cpu_baseline(idx, queries, k, n_sample):
    """The oracle (C restatement of the reference algorithm, all host cores) on a bounded sample."""
    from oracle import c_oracle
    from oracle import ivfpq_oracle as orc
    x = queries[:, :n_sample].cpu().numpy()
    vq = idx.vq_codec.codebook.cpu().numpy()
    pq = idx.pq_codec.codebook.cpu().numpy()
    storage = idx._storage.cpu().numpy()
    is_empty = idx._is_empty.cpu().numpy()
    cs, sz = idx._cell_start.cpu().numpy(), idx._cell_size.cpu().numpy()
    a2i = idx._address2id.cpu().numpy()
    cores = os.cpu_count() or 1
    t0 = time.time()
    vals, ids, adr, cells, npl = orc.search(
        x, vq, pq, storage, is_empty, cs, sz, a2i, k, idx.n_probe, idx.distance,
        use_smart_probing=idx.use_smart_probing,
        scan_fn=lambda *a: c_oracle.scan_topk(*a, n_threads=cores))
    dt = time.time() - t0
    return {"value": n_sample / dt, "unit": "queries/s", "cores": cores, "kind": "port",
            "sample": f"{n_sample} of the {queries.shape[1]} queries, full pipeline "
                      f"(numpy coarse+LUT, C/OpenMP list scan), {dt:.1f} s"}, ids


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nq", type=int, default=10000)
    ap.add_argument("--n-base", type=int, default=1000000)
    ap.add_argument("--n-train", type=int, default=100000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--n-cells", type=int, default=1024)
    ap.add_argument("--n-probe", type=int, default=32)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--layout", choices=["packed", "ref"], default="packed")
    ap.add_argument("--cpu-sample", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # TPQ_BENCH_ONE_DEVICE=1 is a validation hook for 1-GPU boxes: every rank uses cuda:0 and the
    # rendezvous runs over gloo (RCCL refuses two ranks on one device); never set by the driver
    one_device = os.environ.get("TPQ_BENCH_ONE_DEVICE", "0") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_device:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    from torchpq_amd import distributed as tpd
    from torchpq_amd.index import IVFPQIndex

    t_train = t_add = 0.0
    if rank == 0:
        idx, base, centers, gen, t_train, t_add = build_index(args, device)
    else:
        idx = IVFPQIndex(d_vector=args.d, n_subvectors=args.m, n_cells=args.n_cells, device=str(device))
        base = None
    idx.n_probe = args.n_probe
    idx.use_smart_probing = False
    idx.use_packed_layout = args.layout == "packed"
    if world > 1:
        tpd.replicate_index(idx, src=0)  # the one collective: RCCL broadcast at load
    # every rank searches its own query set (weak scaling), same distribution, rank-specific seed
    qgen = torch.Generator(device=device)
    qgen.manual_seed(4321 + rank)
    cgen = torch.Generator(device=device)
    cgen.manual_seed(1234)
    centers = make_centers(cgen, args.d, device)
    queries = sift_like(qgen, args.d, args.nq, centers, device)

    scan = idx._ivfpq_topk._scan
    for _ in range(args.warmup):
        idx.search(queries, k=args.k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    scan.record_events = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        vals, ids = idx.search(queries, k=args.k)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    events = scan.record_events
    scan.record_events = None
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- roofline of the dominant kernel (the list scan) --------------------------------------
    scan_ms = float(np.mean([a.elapsed_time(b) for a, b in events])) if events else float("nan")
    _, cells, npl = idx.probe(queries)
    sizes = idx._cell_size[cells]
    live = torch.arange(cells.shape[1], device=device)[None, :] < npl[:, None]
    scanned_slots = int((sizes * live).sum().item())
    algo_bytes = scanned_slots * args.m  # uint8 codes only: the irreducible read
    achieved = algo_bytes / (scan_ms * 1e-3) / 1e9
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                "kernel": "scan_packed_kernel" if (args.layout == "packed") else "scan_ref_kernel",
                "kernel_ms": round(scan_ms, 4), "algorithmic_bytes_per_launch": algo_bytes,
                "bytes_per_query": round(algo_bytes / args.nq, 1),
                "cell_imbalance": round(float((idx._cell_size.double() ** 2).sum().item()) * args.n_cells
                                        / float(idx._cell_size.sum().item()) ** 2, 3)}

    # HBM-side bytes per launch come from a separate rocprofv3 --pmc pass over this same command
    # (FETCH_SIZE, corrected as MI355X_MICROARCH.md prescribes); the committed summary is read back
    prof = os.path.join(ROOT, "profiles", "r01_bench_scan_packed.json")
    if args.layout == "packed" and os.path.exists(prof):
        try:
            pj = json.load(open(prof))
            if abs(pj["algorithmic_bytes_per_launch"] - algo_bytes) <= 0.02 * algo_bytes:
                roofline["traffic"] = round(pj["hbm_side_read_bytes_corrected"])
                roofline["traffic_source"] = "profiles/r01_bench_scan_packed.json (rocprofv3 --pmc FETCH_SIZE x2)"
        except Exception:
            pass

    out = {
        "metric": "queries/sec + recall@100, SIFT1M IVFPQ d=128 m=64 nprobe=32",
        "value": round(args.nq * args.steps * world / dt, 1), "unit": "queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32",
        "data": "synthetic (SIFT1M-shaped: non-negative integer-valued clustered fp32)",
        "config": {"workload": f"SIFT1M-like d={args.d} n={args.n_base} IVFPQ n_cells={args.n_cells} "
                               f"m={args.m} nprobe={args.n_probe} k={args.k} on 1xMI355X per rank",
                   "n_query_per_rank": args.nq, "code_layout": args.layout,
                   "codes": "u8 (8-bit PQ)", "arithmetic": "f32 LUT entries, f32 sums, exact ids",
                   "use_smart_probing": False, "parallelism": f"query-sharded x{world}, replicated index"},
        "roofline": roofline,
    }
    if rank == 0:
        out["train_s"] = round(t_train, 2)
        out["add_s"] = round(t_add, 2)
        # recall@100 against exact search on the raw vectors (true nearest neighbour in the top-k,
        # the reference benchmark's definition -- BASELINE.md) on a 1000-query sample
        ns = min(1000, args.nq)
        d2 = (-2.0 * queries[:, :ns].T @ base) + (base * base).sum(0)[None, :]
        nn = d2.argmin(dim=1)
        out["recall_gt@%d" % args.k] = round(float((ids[:ns] == nn[:, None]).any(dim=1).float().mean().item()), 4)
        if not args.no_cpu_baseline and world == 1:  # CPU baseline: rank 0 at N=1 only
            cb, cpu_ids = cpu_baseline(idx, queries, args.k, min(args.cpu_sample, args.nq))
            out["cpu_baseline"] = cb
            gpu_ids = ids[:cpu_ids.shape[0]].cpu().numpy()
            inter = [len(np.intersect1d(gpu_ids[q], cpu_ids[q])) for q in range(cpu_ids.shape[0])]
            out["recall_vs_ref@%d" % args.k] = round(float(np.mean(inter)) / args.k, 4)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
