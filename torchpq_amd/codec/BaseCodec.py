"""Common base of the coarse (VQ) and product (PQ) quantisers.

State-dict contract shared with the reference (torchpq/codec/BaseCodec.py:5-28): one boolean
buffer ``_is_trained`` next to the k-means child module, so trained indexes interchange."""
import torch

from ..CustomModule import CustomModule
from ..util import tensor_version


class BaseCodec(CustomModule):
    """Sub-classes provide train(x), encode(x) and decode(code)."""

    def __init__(self):
        super().__init__()
        self.register_buffer("_is_trained", torch.zeros((), dtype=torch.bool))

    @property
    def is_trained(self):
        # bool() of a device tensor is a host sync: read the flag once per buffer state (train(),
        # load_state_dict() and .to() all install a NEW tensor), so search() stays sync-free and
        # can be captured in a HIP graph
        # (in-place writes -- `_is_trained.fill_()`, `.data = ...`, a stock load_state_dict
        # copying into the buffer -- bump the tensor's version counter or replace its storage)
        t = self._is_trained
        key = (tensor_version(t), t.data_ptr())  # inference tensors carry no version counter
        cached = self.__dict__.get("_trained_seen")
        if cached is None or cached[0] is not t or cached[1] != key:
            cached = (t, key, bool(t))
            self.__dict__["_trained_seen"] = cached
        return cached[2]

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
