import sys, os, torch
sys.path.insert(0, "/root/repo")
import bench
from torchpq_amd import kernels as K
dev = torch.device("cuda", 0)
synth = bench.SiftLike(128, dev)
x = synth.sample(10000, seed=4321)
c = synth.sample(4096, seed=7)
z = torch.zeros(4096, device=dev, dtype=torch.long)
ref = K.CoarseProbeHip(route="fp32")(x, c, z, z, 16, None)
got = K.CoarseProbeHip(route="fp16")(x, c, z, z, 16, None)
torch.cuda.synchronize()
for name, a, b in zip(("sims", "cells", "n"), ref, got):
    if a is None: continue
    ne = (a != b)
    print(name, "mismatch elems", int(ne.sum()), "rows", int(ne.reshape(a.shape[0], -1).any(1).sum()) if a.dim() > 1 else int(ne.sum()))
ne = (ref[1] != got[1]).any(1).nonzero().flatten()[:3]
for r in ne.tolist():
    print(r, ref[1][r].tolist(), got[1][r].tolist())
