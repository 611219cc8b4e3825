from .IVFPQIndex import IVFPQIndex
