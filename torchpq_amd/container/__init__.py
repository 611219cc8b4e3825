from .BaseContainer import BaseContainer
from .CellContainer import CellContainer
