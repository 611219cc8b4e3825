"""Coarse-probe similarity (torchpq/metric.py:31-98).

The coarse query x cell-centroid product is a plain library GEMM (rocBLAS/hipBLASLt through
torch.matmul), exactly as the reference uses cuBLAS; the epilogue follows the reference's
order: ``y = a^T b; y *= 2; y -= |a|^2; y -= |b|^2``.
"""


def negative_squared_l2_distance(a, b, inplace=False, use_tensor_core=False, scale_mode="none"):
    """a [.., d, m], b [.., d, n] -> [.., m, n] fp32.  ``use_tensor_core``/``scale_mode`` are the
    reference's fp16 knobs (metric.py:47-73); the fp16 path would break the 1e-4 distance
    tolerance on MI355X just as it does on NVIDIA, so fp32 is always used here."""
    y = a.transpose(-2, -1).contiguous() @ b
    y.mul_(2)
    y.sub_((a * a).sum(dim=-2)[..., :, None])
    y.sub_((b * b).sum(dim=-2)[..., None, :])
    return y


def cosine_similarity(a, b, normalize=True, inplace=False):
    """torchpq/metric.py:4-29 (never mutates its inputs)."""
    if normalize:
        a = a / (a.norm(dim=-2, keepdim=True) + 1e-8)
        b = b / (b.norm(dim=-2, keepdim=True) + 1e-8)
    return a.transpose(-2, -1) @ b
