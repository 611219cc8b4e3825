import importlib.util, os, sys, json, torch, numpy as np
ROOT = "/root/repo"; sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
args = bench.parse_args(["--no-secondary"])
dev = torch.device("cuda:0")
synth = bench.SiftLike(args.d, dev)
base = synth.sample(args.n_base, seed=1)
g = torch.Generator(device=dev); g.manual_seed(2)
train = base[:, torch.randperm(args.n_base, generator=g, device=dev)[:args.n_train]].contiguous()
idx, _, _ = bench.build_index(args, dev, base, train)
idx.n_probe = 32; idx.use_smart_probing = False
q = synth.sample(10000, seed=4321)
_, cells, _ = idx.probe(q)
sizes = idx._cell_size
def pair_stats(order):
    c = cells[order]
    a, b = c[0::2], c[1::2]
    shared = (a[:, :, None] == b[:, None, :])            # [pairs, 32, 32]
    in_b = shared.any(2)                                    # a's cells also in b
    sa = sizes[a]; sb = sizes[b]
    bytes_sep = (sa.sum(1) + sb.sum(1)).double()
    bytes_shared = (sa * in_b).sum(1).double()
    return float(in_b.float().mean()), float((bytes_shared.sum()) / bytes_sep.sum())
print("random pairs", pair_stats(torch.arange(10000, device=dev)))
print("sorted by first cell", pair_stats(torch.argsort(cells[:, 0] * 2048 + cells[:, 1], stable=True)))
# greedy: sort by the sorted tuple of first 3 cells
key = cells[:, 0] * (1 << 22) + cells[:, 1] * (1 << 11) + cells[:, 2]
print("sorted by first three cells", pair_stats(torch.argsort(key)))
