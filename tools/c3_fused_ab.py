#!/usr/bin/env python
"""A/B at the GIST1M shape (d=960, m=120, ds=8): LUT materialised by tpq_adc_lut vs built inside the
scan workgroups; results must be identical."""
import importlib.util, json, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
from torchpq_amd.index import IVFPQIndex
dev = torch.device("cuda:0")
d, m, n_cells, n, n_probe, k = 960, 120, 1024, 1_000_000, 64, 100
g = torch.Generator(device=dev); g.manual_seed(1235)
centers = torch.rand(d, 512, generator=g, device=dev)
def sample(c):
    a = torch.randint(0, 512, (c,), generator=g, device=dev)
    return (centers[:, a] * 0.6 + torch.randn(d, c, generator=g, device=dev) * 0.08).clamp_(0, 1)
np.random.seed(1)
idx = IVFPQIndex(d_vector=d, n_subvectors=m, n_cells=n_cells, initial_size=2 * n // n_cells, device="cuda:0")
idx.train(sample(100_000))
for b in range(0, n, 1 << 17):
    idx.add(sample(min(1 << 17, n - b)))
idx.n_probe, idx.use_smart_probing = n_probe, False
for nq in (1, 100, 1000, 10000):
    q = sample(nq)
    out = {"nq": nq}
    ref = None
    for thr in (4, 8):
        idx.fused_lut_max_subvector = thr
        dt, scan_ms, _, v, i = bench.time_search(idx, q, k, 10, 2)
        out[f"ms_thr{thr}"] = round(dt / 10 * 1e3, 4)
        out[f"scan_ms_thr{thr}"] = round(scan_ms, 4)
        if ref is None:
            ref = (v, i)
        else:
            out["identical"] = bool(torch.equal(ref[0], v) and torch.equal(ref[1], i))
    print(json.dumps(out), flush=True)
