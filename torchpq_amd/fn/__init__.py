"""Kernel-variant dispatchers of the search path (the reference's torchpq/fn package)."""
from .IVFPQTopk import IVFPQTopk  # list scan
from .Topk import Topk            # row-wise top-k

__all__ = ["IVFPQTopk", "Topk"]
