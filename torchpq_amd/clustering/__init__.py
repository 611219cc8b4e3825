"""Lloyd k-means drivers over the fp32-MFMA assign kernel."""
from .KMeans import KMeans
from .MultiKMeans import MultiKMeans

__all__ = ["KMeans", "MultiKMeans"]
