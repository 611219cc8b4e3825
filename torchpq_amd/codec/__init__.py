from .BaseCodec import BaseCodec
from .PQCodec import PQCodec
from .VQCodec import VQCodec
