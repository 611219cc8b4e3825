"""Index classes: the IVFPQ drop-in and the exact (flat) index used as recall ground truth."""
from .FlatIndex import FlatIndex
from .IVFPQIndex import IVFPQIndex

__all__ = ["IVFPQIndex", "FlatIndex"]
