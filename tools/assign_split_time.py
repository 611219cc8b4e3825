#!/usr/bin/env python
"""Time tpq_max_sim_split at the C5 shape (used with TPQ_AMD_LIB variants for knock-out A/Bs)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchpq_amd import kernels as K  # noqa: E402

dev = "cuda:0"
g = torch.Generator(device=dev)
g.manual_seed(0)
L, D, N, KK = 64, int(os.environ.get("D", 64)), 1000000, 256
data = torch.randn(L, D, N, generator=g, device=dev)
cent = data[:, :, :KK].contiguous()
k = K.MaxSimHip(distance="euclidean", precision=os.environ.get("PREC", "bf16x3"))
k(data, cent, dim=2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    k(data, cent, dim=2)
e1.record()
torch.cuda.synchronize()
print(round(e0.elapsed_time(e1) / 10, 3), "ms")
