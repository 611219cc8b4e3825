"""address <-> id bookkeeping (mirrors torchpq/container/BaseContainer.py:8-134)."""
from abc import ABC, abstractmethod

import torch

from .. import util
from ..CustomModule import CustomModule
from ..kernels import GetIdByAddressHip


class BaseContainer(CustomModule, ABC):
    def __init__(self, device="cuda:0", initial_size=None, expand_step_size=1024,
                 expand_mode="double", use_inverse_id_mapping=False):
        super().__init__()
        if initial_size is None:
            initial_size = expand_step_size
        assert expand_mode in ["step", "double"]
        assert initial_size >= 0
        assert expand_step_size > 0
        if torch.device(device).type != "cuda":
            raise RuntimeError(
                "torchpq_amd containers live on an AMD GPU (device='cuda:N'); the reference's "
                "CPU fallbacks are not part of this build")
        self.device = device
        self.device_type = torch.device(device).type
        self.initial_size = initial_size
        self.expand_step_size = expand_step_size
        self.expand_mode = expand_mode
        self.use_inverse_id_mapping = use_inverse_id_mapping
        self._max_id = -1
        self.register_buffer("_address2id",
                             torch.full((initial_size,), -1, device=device, dtype=torch.long))
        self.register_buffer("_id2address", None)
        self._get_id_by_address_hip = GetIdByAddressHip()

    @property
    def capacity(self):
        return self._address2id.shape[0]

    @property
    def max_id(self):
        return self._max_id

    def empty(self):
        self._address2id.fill_(-1)
        self._drop_inverse_id_mapping()

    def _drop_inverse_id_mapping(self):
        del self._id2address
        self.register_buffer("_id2address", None)

    def get_id_by_address(self, address):
        """int64 addresses (any shape) -> ids, -1 where the address is out of range or free."""
        assert util.check_dtype(address, torch.int64)
        address = address.to(self.device)
        return self._get_id_by_address_hip(self._address2id, address)

    def create_inverse_id_mapping(self):
        """_id2address [max_id + 1]: address of every stored id, -1 elsewhere (:100-110)."""
        a2i = self._address2id
        adr = torch.nonzero(a2i >= 0)[:, 0]
        id2a = torch.full((self.max_id + 1,), -1, device=self.device, dtype=torch.long)
        id2a[a2i[adr]] = adr
        del self._id2address
        self.register_buffer("_id2address", id2a)

    def get_address_by_id(self, ids):
        """ids int64 [n] -> addresses, -1 for unknown ids (:79-98).  The inverse table is rebuilt
        after every add/remove (the reference keeps serving a stale one)."""
        assert util.check_dtype(ids, torch.int64)
        ids = ids.to(self.device)
        if self._id2address is None:
            self.create_inverse_id_mapping()
        mask = (0 <= ids) & (ids <= self.max_id)
        address = torch.full_like(ids, -1)
        address[mask] = self._id2address[ids[mask]]
        return address

    def _after_load_state_dict(self):
        # `_max_id` is not part of the reference's state_dict: recover it.
        a2i = self._address2id
        self._max_id = int(a2i.max().item()) if a2i.numel() else -1
        self.device = str(a2i.device)

    @abstractmethod
    def add(self):
        pass

    @abstractmethod
    def remove(self):
        pass
