from .FlatIndex import FlatIndex
from .IVFPQIndex import IVFPQIndex
