"""Inverted-list storage of the IVFPQ index."""
from .BaseContainer import BaseContainer
from .CellContainer import CellContainer

__all__ = ["BaseContainer", "CellContainer"]
