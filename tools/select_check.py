#!/usr/bin/env python
"""tpq_max_sim_select against tpq_max_sim at the C5 shape: labels must be identical; both times."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchpq_amd import kernels as K  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev)
g.manual_seed(0)
L, D, N, KK = int(os.environ.get("L", 64)), int(os.environ.get("D", 64)), int(os.environ.get("N", 1000000)), 256
data = torch.randn(L, D, N, generator=g, device=dev)
cent = data[:, :, :KK].contiguous() + 0.3
ex, se = K.MaxSimHip(), K.MaxSimSelectHip()
ve, le = ex(data, cent, dim=2)
vs, ls = se(data, cent)
print("labels identical", bool(torch.equal(le, ls)), "mismatches", int((le != ls).sum()),
      "max |dv|/|v|", float(((vs - ve).abs() / ve.abs().clamp_min(1e-6)).max()))
from torchpq_amd import _lib  # noqa: E402
for name, fn in (("select", lambda: se(data, cent)), ("fp32", lambda: ex(data, cent, dim=2))):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(name, round(e0.elapsed_time(e1) / 5, 3), "ms")
