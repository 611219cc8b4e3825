import sys, time, torch, importlib.util, os
ROOT="/root/repo"; sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("b100", os.path.join(ROOT, "tools", "build_100m.py"))
tool = importlib.util.module_from_spec(spec); spec.loader.exec_module(tool)
from torchpq_amd.container import CellContainer
idx, cells_all, centers, t = tool.build(40_000_000, 1 << 20)
print("built 40M", {k: round(v, 3) if isinstance(v, float) else v for k, v in t.items()})
x = tool.chunk_vectors(128, 1 << 20, 777, centers, torch.device("cuda:0"))
def T(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3, r
ms, cells = T(lambda: idx.vq_codec.encode(x)); print("vq encode", round(ms, 2))
ms, codes = T(lambda: idx.pq_codec.encode(x)); print("pq encode", round(ms, 2))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
ms, _ = T(lambda: CellContainer.add(idx, codes, cells)); pr.disable()
print("container add", round(ms, 2))
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
ms, _ = T(lambda: torch.empty(16, idx.capacity + 5_000_000, 4, device="cuda", dtype=torch.uint8)); print("alloc new storage", round(ms, 2))
