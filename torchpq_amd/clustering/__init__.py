from .KMeans import KMeans
from .MultiKMeans import MultiKMeans
