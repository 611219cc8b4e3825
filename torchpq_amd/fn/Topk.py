"""Row-wise top-k dispatch (mirrors torchpq/fn/Topk.py:5-67).  The reference picks one of seven
CUDA kernels by k; one wave-per-row HIP kernel family covers k <= 1024 here."""
import torch

from ..kernels import TopkSelectHip


class Topk:
    def __init__(self):
        self._select = TopkSelectHip()

    def __call__(self, x, k=1, dim=1):
        if dim == -1:
            dim = 1
        assert dim == 1, "only support last dimention"
        assert len(x.shape) == 2, "only support 2d tensors"
        assert x.is_contiguous(), "x is not contiguous"
        assert k >= 1
        assert x.device.type == "cuda"
        if k <= 1024:
            return self._select(x, k=k, dim=dim)
        return torch.topk(x, dim=dim, k=k)
