#!/usr/bin/env python
"""How much of its error bound does the fast path of tpq_coarse_assign actually use?  For points the
selection decided (not re-checked), the returned maximum is the FAST value b1 - |a - mu|^2; compare it
with the float64 similarity of the same centroid and express the difference in units of the bound's
fast-path term eps_fast (|a - mu| + |c - mu|max)^2."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from torchpq_amd import kernels as K  # noqa: E402

dev = "cuda:0"
out = {}
for kind in ("clustered", "gauss", "sift"):
    for d, n in ((128, 16384), (64, 1024)):
        m = 100_000
        g = torch.Generator(device=dev)
        g.manual_seed(d + n)
        if kind == "gauss":
            A = torch.randn(d, m, generator=g, device=dev) * 10
            B = A[:, torch.randperm(m, generator=g, device=dev)[:n]].contiguous() + 0.1
        else:
            centers = torch.randn(d, 4096, generator=g, device=dev).abs() * 45.0
            A = centers[:, torch.randint(0, 4096, (m,), generator=g, device=dev)] + \
                torch.randn(d, m, generator=g, device=dev) * 12.0
            if kind == "sift":
                A = A.clamp_(0, 255).round_()
            A = A.contiguous()
            B = A[:, torch.randperm(m, generator=g, device=dev)[:n]].contiguous() + 0.5
        op = K.CoarseAssignHip()
        v, i = op(A, B, return_vals=True)
        a64, b64 = A.double(), B.double()
        mu = b64.mean(1, keepdim=True)
        sel = b64[:, i]                                            # [d, m] chosen centroid per point
        true = -((a64 - sel) ** 2).sum(0)
        an = (a64 - mu).norm(dim=0)
        cn = (b64 - mu).norm(dim=0).max()
        ks = 8 if d > 64 else 4
        eps_fast = 3.03 / 65536 + (16 * ks + 5 + 8) / 8388608
        used = ((v.double() - true).abs() / (eps_fast * (an + cn) ** 2))
        out[f"{kind}_{d}x{n}"] = {"max_share_of_bound": round(float(used.max()), 4),
                                  "mean_share_of_bound": round(float(used.mean()), 5),
                                  "rechecked_share": round(op.last_rechecked() / m, 4)}
print(json.dumps(out, indent=1))
