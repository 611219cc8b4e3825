import sys, json, torch
sys.path.insert(0, "/root/repo")
from torchpq_amd import kernels as K
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(0)
for (d, n, k) in [(128, 1 << 20, 1024), (128, 1 << 20, 16384), (96, 1 << 20, 4096), (128, 100000, 1024), (128, 1 << 20, 512)]:
    data = torch.randn(1, d, n, generator=g, device=dev)
    cent = torch.randn(1, d, k, generator=g, device=dev)
    ms = K.MaxSimHip(distance="euclidean")
    v, i = ms(data, cent, dim=2, mode="tn")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): ms(data, cent, dim=2, mode="tn")
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 5
    print(json.dumps({"d": d, "n": n, "k": k, "ms": round(t, 3), "TFLOPs": round(2.0 * n * k * d / t / 1e9, 1), "chk": int(i.sum().item()) % 100003}))
