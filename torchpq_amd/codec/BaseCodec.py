"""Common base of the coarse (VQ) and product (PQ) quantisers.

State-dict contract shared with the reference (torchpq/codec/BaseCodec.py:5-28): one boolean
buffer ``_is_trained`` next to the k-means child module, so trained indexes interchange."""
import torch

from ..CustomModule import CustomModule


class BaseCodec(CustomModule):
    """Sub-classes provide train(x), encode(x) and decode(code)."""

    def __init__(self):
        super().__init__()
        self.register_buffer("_is_trained", torch.zeros((), dtype=torch.bool))

    @property
    def is_trained(self):
        return bool(self._is_trained)

    def _trained(self, value):
        if not isinstance(value, bool):
            raise AssertionError("trained flag must be a bool")
        self._is_trained = torch.tensor(value, device=self._is_trained.device)

    def _check_trained(self, what="codec"):
        assert self.is_trained, f"{what} is not trained"

    def train(self, *args, **kwargs):  # pragma: no cover - interface
        raise NotImplementedError

    def encode(self, *args, **kwargs):  # pragma: no cover - interface
        raise NotImplementedError

    def decode(self, *args, **kwargs):  # pragma: no cover - interface
        raise NotImplementedError
