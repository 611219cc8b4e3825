"""torchpq/codec/BaseCodec.py:5-28."""
from abc import ABC, abstractmethod

import torch

from ..CustomModule import CustomModule


class BaseCodec(CustomModule, ABC):
    def __init__(self):
        super().__init__()
        self.register_buffer("_is_trained", torch.tensor(False))

    def _trained(self, value):
        assert type(value) == bool
        self._is_trained.data = torch.tensor(value)

    @property
    def is_trained(self):
        return bool(self._is_trained.item())

    @abstractmethod
    def train(self):
        pass

    @abstractmethod
    def encode(self):
        pass

    @abstractmethod
    def decode(self):
        pass
