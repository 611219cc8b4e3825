"""Exact (brute-force) search: drop-in for torchpq.index.FlatIndex (reference index/FlatIndex.py:8-101
over container/FlatContainer.py).  SURVEY 8(f) rank 4: the ground-truth generator for recall.

search = one library GEMM (rocBLAS, as the reference uses cuBLAS) + the HIP row top-k select
(tpq_topk_select) + address->id.  Storage is the reference's dense `_storage [d, capacity, 1]`
with `_address2id`; vectors are appended, removed slots are tombstoned (id -1) and reused.
"""
import torch

from .. import metric, util
from ..container.BaseContainer import BaseContainer
from ..fn import Topk


class FlatIndex(BaseContainer):
    def __init__(self, d_vector, initial_size=None, expand_step_size=1024, expand_mode="double",
                 device="cuda:0", distance="euclidean", verbose=0):
        super().__init__(device=device, initial_size=initial_size, expand_step_size=expand_step_size,
                         expand_mode=expand_mode, use_inverse_id_mapping=True)
        self.d_vector = d_vector
        self.code_size = d_vector
        self.contiguous_size = 1
        self.dtype = torch.float32
        self.verbose = verbose
        if distance in ["euclidean", "l2"]:
            self.distance = "euclidean"
        elif distance in ["cosine", "angular"]:
            self.distance = "cosine"
        elif distance in ["inner", "dot"]:
            self.distance = "inner"
        elif distance in ["manhattan", "l1"]:
            raise NotImplementedError("currently manhattan distance is not supported")
        else:
            raise NotImplementedError(f"unknown distance metric: {distance}")
        self._n_items = 0
        self.register_buffer("_storage", torch.zeros(d_vector, self.initial_size, 1, device=device,
                                                     dtype=torch.float32))
        self._topk = Topk()

    @property
    def n_items(self):
        return self._n_items

    def _after_load_state_dict(self):
        super()._after_load_state_dict()
        self._n_items = int((self._address2id >= 0).sum().item())

    def _grow_to(self, needed):
        cap = self.capacity
        if needed <= cap:
            return
        new_cap = cap
        step = self.expand_step_size
        while new_cap < needed:
            new_cap = max(new_cap * 2, 1) if self.expand_mode == "double" else new_cap + step
        storage = torch.zeros(self.d_vector, new_cap, 1, device=self.device, dtype=torch.float32)
        storage[:, :cap] = self._storage
        a2i = torch.full((new_cap,), -1, device=self.device, dtype=torch.long)
        a2i[:cap] = self._address2id
        del self._storage, self._address2id
        self.register_buffer("_storage", storage)
        self.register_buffer("_address2id", a2i)

    def add(self, x, ids=None, return_address=False):
        """x [d_vector, n] f32; ids default arange + max_id + 1; free slots are filled in order."""
        assert len(x.shape) == 2 and x.shape[0] == self.d_vector
        assert util.check_dtype(x, "float32")
        x = x.to(self.device)
        n = x.shape[1]
        if ids is None:
            ids = torch.arange(n, device=self.device, dtype=torch.long) + self.max_id + 1
        else:
            assert util.check_dtype(ids, torch.int64) and ids.shape[0] == n
            ids = ids.to(self.device)
        if n == 0:
            return (ids, ids.clone()) if return_address else ids
        self._grow_to(self._n_items + n)
        free = torch.nonzero(self._address2id < 0)[:n, 0]
        self._storage[:, free, 0] = x
        self._address2id[free] = ids
        self._max_id = max(self._max_id, ids.max().item())
        self._n_items += n
        self._drop_inverse_id_mapping()
        return (ids, free) if return_address else ids

    def remove(self, ids=None, address=None):
        if ids is not None:
            address = self.get_address_by_id(ids)
        elif address is None:
            raise RuntimeError("Need either ids or address")
        address = address.to(self.device)
        address = address[(address >= 0) & (address < self.capacity)].unique()
        address = address[self._address2id[address] >= 0]
        if address.shape[0] == 0:
            return
        self._address2id[address] = -1
        self._storage[:, address] = 0
        self._n_items -= address.shape[0]
        self._drop_inverse_id_mapping()

    def search(self, x, k=1, return_address=False):
        """x [d_vector, n_query] f32 -> (values [n_query, k] descending, ids[, address])"""
        d_vector, n_query = x.shape
        assert d_vector == self.d_vector
        assert util.check_dtype(x, "float32")
        assert k >= 1
        x = x.to(self.device)
        storage = self._storage.view(self.d_vector, -1)
        if self.distance == "euclidean":
            sims = metric.negative_squared_l2_distance(x, storage)
        elif self.distance == "cosine":
            sims = metric.cosine_similarity(x, storage, normalize=True)
        else:
            sims = metric.cosine_similarity(x, storage, normalize=False)
        sims = sims.masked_fill((self._address2id < 0)[None, :], float("-inf")).contiguous()
        topk_val, topk_address = self._topk(sims, k=min(k, sims.shape[1]), dim=1)
        topk_address = torch.where(torch.isneginf(topk_val), torch.full_like(topk_address, -1), topk_address)
        topk_ids = self.get_id_by_address(topk_address)
        if return_address:
            return topk_val, topk_ids, topk_address
        return topk_val, topk_ids
