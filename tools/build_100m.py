#!/usr/bin/env python
"""SURVEY 8 row f-2 at size: build a 100 M-vector IVFPQ index (BASELINE.json configs[3]: d=128,
n_cells=16384, m=64) THROUGH IVFPQIndex.train / add, in chunks of generated vectors, and check the
placement against the sort-by-cell oracle (for an initially empty index the i-th vector assigned
to cell c sits at cell_start[c] + i, input order -- CellContainer.py:313-367, get_ioa.cu:9-47,
get_write_address_v2.cu:9-41) on sampled cells, and the stored codes against encode() on sampled
chunks.

    python tools/build_100m.py [--n 100000000] [--chunk 1048576] [--out profiles/r03_build_100m.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def chunk_vectors(d, count, seed, centers, device):
    """chunk `seed` of the synthetic base set: clustered, non-negative, integer-valued"""
    g = torch.Generator(device=device)
    g.manual_seed(1000 + seed)
    a = torch.randint(0, centers.shape[1], (count,), generator=g, device=device)
    x = centers[:, a] + torch.randn(d, count, generator=g, device=device) * 25.0
    return x.abs_().round_().clamp_(0, 218)


def build(n_total=100_000_000, chunk=1 << 20, d=128, m=64, n_cells=16384, n_train=1_000_000,
          device="cuda:0", initial_size=None, verbose=False):
    """returns (index, cell of every vector [n_total] int16/int32, centers, timings dict)"""
    from torchpq_amd.index import IVFPQIndex
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(99)
    centers = torch.randn(d, 4096, generator=g, device=dev).abs() * 45.0
    np.random.seed(99)
    idx = IVFPQIndex(d_vector=d, n_subvectors=m, n_cells=n_cells, initial_size=initial_size,
                     device=device)
    t = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    idx.train(chunk_vectors(d, n_train, -1, centers, dev))
    torch.cuda.synchronize()
    t["train_s"] = time.perf_counter() - t0
    cells_all = torch.empty(n_total, device=dev, dtype=torch.int16 if n_cells <= 32768 else torch.int32)
    t_gen = t_add = 0.0
    grows = 0
    for ci, b in enumerate(range(0, n_total, chunk)):
        cnt = min(chunk, n_total - b)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        x = chunk_vectors(d, cnt, ci, centers, dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        cap_before = idx.capacity
        ids, adr = idx.add(x, return_address=True)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        t_gen += t1 - t0
        t_add += t2 - t1
        grows += int(idx.capacity != cap_before)
        cells_all[b:b + cnt] = idx.get_cell_by_address(adr).to(cells_all.dtype)
        if verbose and ci % 10 == 0:
            print(f"chunk {ci}: {cnt} vectors, add {1e3 * (t2 - t1):.1f} ms, capacity {idx.capacity}", flush=True)
    idx.release_spare()  # the idle growth arena: as large as the index
    t.update({"generate_s": t_gen, "add_s": t_add, "chunks": (n_total + chunk - 1) // chunk,
              "chunks_that_grew_the_storage": grows, "n": n_total, "capacity": idx.capacity,
              "vectors_per_s": n_total / t_add})
    return idx, cells_all, centers, t


def check_placement(idx, cells_all, n_sample_cells=48, seed=0):
    """sampled cells: slot order == insertion (= id) order of the vectors assigned to the cell"""
    dev = cells_all.device
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    sample = torch.randperm(idx.n_cells, generator=g, device=dev)[:n_sample_cells]
    sizes = torch.bincount(cells_all.long(), minlength=idx.n_cells)
    assert torch.equal(sizes, idx._cell_size), "cell sizes differ from the assignment histogram"
    st, sz, cap = idx._cell_start, idx._cell_size, idx._cell_capacity
    assert torch.equal(st, torch.cumsum(cap, 0) - cap) and bool((sz <= cap).all())
    checked = 0
    for c in sample.tolist():
        want = torch.nonzero(cells_all == c)[:, 0]                       # ascending id = input order
        s0, n_c, cp = int(st[c]), int(sz[c]), int(cap[c])
        got = idx._address2id[s0:s0 + n_c]
        assert torch.equal(got, want), f"cell {c}: placement differs from the sort-by-cell oracle"
        assert bool((idx._is_empty[s0:s0 + n_c] == 0).all()) and bool((idx._is_empty[s0 + n_c:s0 + cp] == 1).all())
        assert bool((idx._address2id[s0 + n_c:s0 + cp] == -1).all())
        checked += n_c
    return checked


def check_codes(idx, centers, chunk, chunk_ids=(0, 37, 95), n_total=None):
    """stored codes of whole chunks == encode() of the regenerated vectors"""
    dev = centers.device
    checked = 0
    for ci in chunk_ids:
        b = ci * chunk
        if n_total is not None and b >= n_total:
            continue
        cnt = min(chunk, (n_total or b + chunk) - b)
        x = chunk_vectors(idx.d_vector, cnt, ci, centers, dev)
        ids = torch.arange(b, b + cnt, device=dev)
        adr = idx.get_address_by_id(ids)
        assert bool((adr >= 0).all())
        assert torch.equal(idx.get_data_by_address(adr), idx.encode(x)), f"chunk {ci}: stored codes != encode"
        assert torch.equal(idx.get_cell_by_address(adr), idx.vq_codec.encode(x))
        checked += cnt
    return checked


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=100_000_000)
    ap.add_argument("--chunk", type=int, default=1 << 20)
    ap.add_argument("--out", default=None)
    ap.add_argument("--verbose", action="store_true")
    a = ap.parse_args()
    t0 = time.time()
    idx, cells_all, centers, t = build(a.n, a.chunk, verbose=a.verbose)
    t["placement_slots_checked"] = check_placement(idx, cells_all)
    t["codes_checked"] = check_codes(idx, centers, a.chunk, n_total=a.n)
    # one search over the finished index (scan layout built lazily here)
    idx.n_probe, idx.use_smart_probing = 64, False
    q = chunk_vectors(idx.d_vector, 10000, 12345, centers, centers.device)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    v, i = idx.search(q, k=100)
    torch.cuda.synchronize()
    t["first_search_s_incl_scan_layout_build"] = time.perf_counter() - t1
    t1 = time.perf_counter()
    v, i = idx.search(q, k=100)
    torch.cuda.synchronize()
    t["search_10k_queries_ms"] = (time.perf_counter() - t1) * 1e3
    t["wall_s"] = time.time() - t0
    t["workload"] = (f"d=128 n={a.n} IVFPQ n_cells=16384 m=64, train on 1 M, add in chunks of {a.chunk} "
                     "generated vectors (default initial_size: cells grow by doubling)")
    t = {k: (round(v, 4) if isinstance(v, float) else v) for k, v in t.items()}
    print(json.dumps(t))
    if a.out:
        with open(a.out, "w") as f:
            f.write(json.dumps(t, indent=1) + "\n")


if __name__ == "__main__":
    main()
