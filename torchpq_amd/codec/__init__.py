"""Quantisers: coarse inverted-file VQ and 8-bit product quantiser."""
from .BaseCodec import BaseCodec
from .PQCodec import PQCodec
from .VQCodec import VQCodec

__all__ = ["BaseCodec", "PQCodec", "VQCodec"]
