"""address <-> id bookkeeping (mirrors torchpq/container/BaseContainer.py:8-134)."""
from abc import ABC, abstractmethod

import torch

from .. import util
from ..CustomModule import CustomModule
from ..kernels import GetAddressByIdHip, GetIdByAddressHip


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
        self._sparse_id_map = None  # (sorted ids, their addresses) when ids are too sparse for a table
        self._get_id_by_address_hip = GetIdByAddressHip()
        self._get_address_by_id_hip = GetAddressByIdHip()

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
        self._sparse_id_map = None

    def get_id_by_address(self, address):
        """int64 addresses (any shape) -> ids, -1 where the address is out of range or free."""
        assert util.check_dtype(address, torch.int64)
        address = address.to(self.device)
        return self._get_id_by_address_hip(self._address2id, address)

    def create_inverse_id_mapping(self):
        """_id2address [max_id + 1]: address of every stored id, -1 elsewhere (:100-110).
        The reference's dense table needs max_id + 1 entries whatever the number of items (its own
        tests draw ids below 2**62, tests/CellContainerTestCase.py:60-66: 32 EiB); ids much sparser
        than the capacity are served from a sorted (id, address) list by binary search instead."""
        a2i = self._address2id
        adr = torch.nonzero(a2i >= 0)[:, 0]
        del self._id2address
        if self.max_id + 1 > 8 * self.capacity + (1 << 20):
            ids, order = torch.sort(a2i[adr])
            self.register_buffer("_id2address", None)
            self._sparse_id_map = (ids, adr[order])
            return
        id2a = torch.full((self.max_id + 1,), -1, device=self.device, dtype=torch.long)
        id2a[a2i[adr]] = adr
        self.register_buffer("_id2address", id2a)

    def get_address_by_id(self, ids):
        """ids int64 [n] -> addresses, -1 for unknown ids (:79-98).  The inverse table is rebuilt
        after every add/remove (the reference keeps serving a stale one)."""
        assert util.check_dtype(ids, torch.int64)
        ids = ids.to(self.device)
        if not self.use_inverse_id_mapping:
            # the reference's linear search (kernels/cuda/get_address_by_id.cu:8-44): no table kept
            return self._get_address_by_id_hip(self._address2id, ids.reshape(-1)).reshape(ids.shape)
        if self._id2address is None and self._sparse_id_map is None:
            self.create_inverse_id_mapping()
        if self._sparse_id_map is not None:
            known, known_adr = self._sparse_id_map
            if known.numel() == 0:
                return torch.full_like(ids, -1)
            pos = torch.searchsorted(known, ids).clamp_(max=known.numel() - 1)
            return torch.where(known[pos] == ids, known_adr[pos], torch.full_like(ids, -1))
        mask = (0 <= ids) & (ids <= self.max_id)
        address = torch.full_like(ids, -1)
        address[mask] = self._id2address[ids[mask]]
        return address

    def _after_load_state_dict(self):
        # `_max_id` is not part of the reference's state_dict: recover it.
        a2i = self._address2id
        self._max_id = int(a2i.max().item()) if a2i.numel() else -1
        self.device = str(a2i.device)

    def expand(self):
        """Grow the id table by one step (BaseContainer.py:112-127): `expand_step_size` doubles first
        in "double" mode; new addresses are free (-1).  Subclasses that also own storage override it
        (CellContainer.expand grows per cell)."""
        if self.expand_mode == "double":
            self.expand_step_size *= 2
        a2i = torch.cat([self._address2id, torch.full(
            (self.expand_step_size,), -1, device=self._address2id.device, dtype=torch.long)])
        del self._address2id
        self.register_buffer("_address2id", a2i)

    @abstractmethod
    def add(self):
        pass

    @abstractmethod
    def remove(self):
        pass
