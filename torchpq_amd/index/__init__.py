"""Index classes: the IVFPQ drop-in, the exact (flat) index used as recall ground truth, and the
HIP-graph replay helper for fixed-shape serving batches."""
from . import FlatIndex as _flat_module
from . import IVFPQIndex as _ivfpq_module
from . import graphed as _graphed_module

IVFPQIndex = _ivfpq_module.IVFPQIndex
FlatIndex = _flat_module.FlatIndex
GraphedSearch = _graphed_module.GraphedSearch

__all__ = ["IVFPQIndex", "FlatIndex", "GraphedSearch"]
