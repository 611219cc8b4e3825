"""Inverted-list storage (mirrors torchpq/container/CellContainer.py:10-393).

Buffers and layouts are the reference's (state_dicts interchange):
  _storage [code_size/4, capacity, 4] u8, _cell_start/_cell_size/_cell_capacity [n_cells] i64,
  _is_empty [capacity] u8 (1 = free), _address2id [capacity] i64 (-1 = none).
On top of them a derived, NON-persistent `_packed` buffer holds the same codes in the MI355X
scan layout (csrc/scan_layout.h); it is kept in step by add() and rebuilt lazily otherwise.
"""
import torch

from .. import util
from ..kernels import (GetCellByAddressHip, GetIOAHip, GetWriteAddressHip, GrowCellsHip,
                       PackCodesHip, ScatterCodesHip)
from .BaseContainer import BaseContainer


class CellContainer(BaseContainer):
    def __init__(self, code_size, n_cells, dtype="uint8", device="cuda:0", initial_size=None,
                 expand_step_size=1024, expand_mode="double", use_inverse_id_mapping=False,
                 contiguous_size=1, verbose=0):
        if initial_size is None:
            initial_size = expand_step_size
        super().__init__(device=device, initial_size=initial_size * n_cells,
                         expand_step_size=expand_step_size, expand_mode=expand_mode,
                         use_inverse_id_mapping=use_inverse_id_mapping)
        assert n_cells > 0
        assert code_size > 0
        assert code_size % contiguous_size == 0
        if type(dtype) == str:
            dtype = util.str2dtype(dtype)
        assert dtype == torch.uint8 and contiguous_size == 4, \
            "the IVFPQ path stores uint8 codes with contiguous_size=4 (IVFPQIndex.py:33-42)"
        self.n_cells = n_cells
        self.code_size = code_size
        self.dtype = dtype
        self.contiguous_size = contiguous_size
        self.initial_size = initial_size
        self.verbose = verbose
        cap = n_cells * initial_size
        self.register_buffer("_storage", torch.zeros(code_size // contiguous_size, cap,
                                                     contiguous_size, device=device, dtype=dtype))
        self.register_buffer("_cell_start",
                             torch.arange(n_cells, device=device, dtype=torch.long) * initial_size)
        self.register_buffer("_cell_size", torch.zeros(n_cells, device=device, dtype=torch.long))
        self.register_buffer("_cell_capacity",
                             torch.full((n_cells,), initial_size, device=device, dtype=torch.long))
        self.register_buffer("_is_empty", torch.ones(cap, device=device, dtype=torch.uint8))
        self._packed = None          # scan-layout copy of _storage (derived, never saved)
        self._packed_valid = False
        # Growth arenas.  _grow moves every cell into NEW buffers; allocated afresh each time, a bulk
        # build leaves one dead multi-GB block per add() in the caching allocator (no later request
        # fits an earlier block), and once they fill the HBM the allocator stops to hipFree them
        # all: 5.4 s of a 14 s build of 100 M vectors.  Large buffers are therefore carved from two
        # flat arenas per buffer, sized geometrically, that swap roles on every growth: a few
        # allocations per build instead of one per add().
        self._arena = {}             # name -> [flat tensor or None, flat tensor or None]
        self._arena_side = 0         # side holding the live buffers
        self._codes_version = 0      # bumped whenever codes or their placement change
        self._has_holes = False      # a tombstone inside some [start, start+size)
        self._get_cell_by_address_hip = GetCellByAddressHip()
        self._get_ioa_hip = GetIOAHip()
        self._get_write_address_hip = GetWriteAddressHip()
        self._scatter_codes_hip = ScatterCodesHip()
        self._pack_codes_hip = PackCodesHip()
        self._grow_cells_hip = GrowCellsHip()

    # ---- derived state -------------------------------------------------------------------------
    @property
    def n_items(self):
        return self._cell_size.sum().item()

    def packed_storage(self):
        """The scan-layout codes, (re)built if stale."""
        if self._packed is None or self._packed.shape[1] != self._storage.shape[1]:
            self._packed = None
            self._packed_valid = False
        if not self._packed_valid:
            self._packed = self._pack_codes_hip(self._storage, self._packed)
            self._packed_valid = True
        return self._packed

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        """arena-backed buffers are prefixes of larger allocations: hand out exact-size copies, so a
        saved state_dict carries the reference's shapes AND sizes (torch.save writes the whole
        underlying allocation of a view)"""
        super()._save_to_state_dict(destination, prefix, keep_vars)
        for name in ("_storage", "_address2id", "_is_empty"):
            t = destination.get(prefix + name)
            if t is not None and t.untyped_storage().nbytes() > t.numel() * t.element_size():
                destination[prefix + name] = t.clone()

    def _after_load_state_dict(self):
        super()._after_load_state_dict()
        self._arena = {}
        self._packed = None
        self._packed_valid = False
        self._codes_version += 1
        self._drop_inverse_id_mapping()
        # a foreign state_dict may carry tombstones inside a cell's occupied range
        pos = torch.arange(self.capacity, device=self._is_empty.device)
        cell = torch.repeat_interleave(torch.arange(self.n_cells, device=pos.device),
                                       self._cell_capacity)
        if cell.numel() == pos.numel():
            inside = pos < (self._cell_start + self._cell_size)[cell]
            self._has_holes = bool((inside & (self._is_empty == 1)).any().item())
        else:
            self._has_holes = True

    # ---- look-ups (reference :97-239) ------------------------------------------------------------
    def get_cell_by_address(self, address):
        assert util.check_dtype(address, torch.int64)
        address = address.to(self.device)
        return self._get_cell_by_address_hip(address, self._cell_start,
                                             self._cell_start + self._cell_capacity)

    def get_ioa(self, cells, unique_cells=None):
        assert util.check_dtype(cells, torch.int64)
        return self._get_ioa_hip(cells.to(self.device), n_cells=self.n_cells)

    def get_write_address(self, cells, empty_adr=None, ioa=None):
        assert util.check_dtype(cells, torch.int64)
        cells = cells.to(self.device)
        if ioa is None:
            ioa = self.get_ioa(cells)
        return self._get_write_address_hip(self._is_empty, self._cell_start, self._cell_capacity,
                                           cells, ioa.to(self.device))

    def get_data_by_address(self, address):
        """[n] int64 -> codes [code_size, n]; invalid addresses give zero columns (:151-211)."""
        assert util.check_dtype(address, torch.int64)
        address = address.to(self.device)
        mask = (0 <= address) & (address < self.capacity)
        data = self._storage.index_select(1, torch.where(mask, address, torch.zeros_like(address)))
        data = data * mask[None, :, None].to(data.dtype)
        return data.transpose(1, 2).reshape(self.code_size, -1)

    def set_data_by_address(self, data, address):
        """codes [code_size, n] -> slots `address` (out-of-range addresses are skipped)."""
        assert util.check_dtype(address, torch.int64)
        assert util.check_dtype(data, self.dtype)
        assert data.shape[0] == self.code_size
        assert data.shape[1] == address.shape[0]
        packed = self._packed if self._packed_valid else None
        self._scatter_codes_hip(data.to(self.device), address.to(self.device), self._storage, packed)
        self._codes_version += 1

    def empty(self):
        super().empty()
        self._storage.fill_(0)
        self._cell_size.fill_(0)
        self._is_empty.fill_(1)
        self._packed_valid = False
        self._codes_version += 1
        self._has_holes = False
        self.print_message("index has been emptied", 2)

    # ---- growth ------------------------------------------------------------------------------
    def _grow(self, new_capacity):
        """Re-lay the buffers out for per-cell capacities `new_capacity` (>= current).  Same final
        layout as the reference's per-cell torch.cat loop (:249-311) -- each cell keeps its slots
        and gains free ones at its end -- in one kernel pass (tpq_grow_cells) instead of
        O(capacity) per cell."""
        old_cap = self._cell_capacity
        if bool((new_capacity == old_cap).all().item()):
            return 0
        new_start = torch.cumsum(new_capacity, 0) - new_capacity
        total = int(new_capacity.sum().item())
        g = self._storage.shape[0]
        out = None
        if total * self.code_size >= self.arena_min_bytes:
            other = 1 - self._arena_side
            out = (self._arena_take("storage", other, g * total * 4, torch.uint8).view(g, total, 4),
                   self._arena_take("a2i", other, total, torch.int64),
                   self._arena_take("empty", other, total, torch.uint8))
            self._arena_side = other
        else:
            self._arena = {}
        storage, a2i, is_empty = self._grow_cells_hip(
            self._storage, self._address2id, self._is_empty, self._cell_start, old_cap.contiguous(),
            new_start.contiguous(), new_capacity.contiguous(), total, out=out)
        added = total - self.capacity
        del self._storage, self._address2id, self._is_empty
        self.register_buffer("_storage", storage)
        self.register_buffer("_address2id", a2i)
        self.register_buffer("_is_empty", is_empty)
        self._cell_start.copy_(new_start)
        self._cell_capacity.copy_(new_capacity)
        self._packed = None
        self._packed_valid = False
        self._codes_version += 1
        self._drop_inverse_id_mapping()
        return added

    arena_min_bytes = 64 << 20   # smaller containers allocate exactly (no slack, no second buffer)
    arena_growth = 1.5           # a replaced arena is sized this factor beyond the request

    def _arena_take(self, name, side, numel, dtype):
        """`numel` elements of the arena `name`/`side`, (re)allocated geometrically when too small"""
        pair = self._arena.setdefault(name, [None, None])
        buf = pair[side]
        dev = self._storage.device
        if buf is None or buf.numel() < numel or buf.device != dev:
            pair[side] = buf = None  # release first: the allocator may hand the block back
            buf = torch.empty(int(numel * self.arena_growth), device=dev, dtype=dtype)
            pair[side] = buf
        return buf[:numel]

    def release_spare(self):
        """Free the arena side that holds no live buffer (the destination of the next growth).
        Call it after a bulk build (bench.py, tools/build_100m.py do): the spare is as large as
        the index -- containers above `arena_min_bytes` otherwise keep up to ~3x their size
        resident (two arenas of 1.5x the request).
        Growth INVALIDATES earlier references to `_storage` / `_address2id` / `_is_empty` (and
        tensors of a `state_dict(keep_vars=True)`): the arena side they point into is reused two
        growths later.  `state_dict()` hands out copies and is safe."""
        for pair in self._arena.values():
            pair[1 - self._arena_side] = None

    def expand(self, cells):
        """Grow every cell in `cells` once (double its capacity, or + expand_step_size)."""
        cap = self._cell_capacity.clone()
        cells = cells.to(self.device).unique()
        cap[cells] = cap[cells] * 2 if self.expand_mode == "double" else cap[cells] + self.expand_step_size
        added = self._grow(cap)
        self.print_message(f"Total storage capacity is expanded by {added} for {cells.shape[0]} cells", 2)

    # ---- add / remove --------------------------------------------------------------------------
    def add(self, data, cells, ids=None, return_address=False):
        """data [code_size, n] uint8, cells [n] int64, optional ids [n] int64 (:313-367).
        The i-th vector of the batch assigned to cell c goes to the i-th free slot of c."""
        assert util.check_dtype(data, self.dtype)
        assert util.check_dtype(cells, torch.long)
        assert data.shape[0] == self.code_size
        assert data.shape[1] == cells.shape[0]
        data = data.to(self.device)
        cells = cells.to(self.device).contiguous()
        n_data = data.shape[1]
        if ids is not None:
            assert util.check_dtype(ids, torch.int64)
            assert ids.shape[0] == n_data
            ids = ids.to(self.device)
        else:
            ids = torch.arange(n_data, device=self.device, dtype=torch.int64) + self.max_id + 1
        if n_data == 0:
            return (ids, ids.clone()) if return_address else ids

        unique_cells, counts = cells.unique(return_counts=True)
        ioa = self.get_ioa(cells, unique_cells)
        # grow until every vector has its slot; the reference re-tests after every round of
        # doubling (:338-344) -- the capacities are worked out first, the data moved once
        cap = self._cell_capacity.clone()
        size_c = self._cell_size[cells]
        while True:
            free = cap[cells] - size_c - (ioa + 1)
            need = cells[free < 0].unique()
            if need.shape[0] == 0:
                break
            cap[need] = cap[need] * 2 if self.expand_mode == "double" else cap[need] + self.expand_step_size
        added = self._grow(cap)
        if added:
            self.print_message(f"Total storage capacity is expanded by {added}", 2)

        if self._has_holes:
            write_address = self.get_write_address(cells=cells, ioa=ioa)
        else:
            # every cell is dense in [start, start+size) (add / remove / _grow keep it so; only a
            # foreign state_dict can carry holes): the ioa-th free slot of a cell is simply
            # start + size + ioa -- O(n) instead of a walk over the cell per vector
            # (get_write_address_v2.cu:9-41 scans from the cell start), same addresses
            write_address = self._cell_start[cells] + self._cell_size[cells] + ioa
        self.set_data_by_address(data, write_address)
        self._address2id[write_address] = ids
        self._max_id = max(self._max_id, ids.max().item())
        self._is_empty[write_address] = 0
        self._cell_size[unique_cells] += counts
        self._drop_inverse_id_mapping()
        self.print_message(f"{n_data} new items added", 1)
        return (ids, write_address) if return_address else ids

    def _compact(self):
        """Pack the live slots of every cell to the front of its range (stable), so that
        [start, start+size) holds no tombstone.  Capacities and starts are unchanged.
        Returns the old -> new address map [capacity] (-1 for slots that were empty)."""
        dev = self._storage.device
        cap = self.capacity
        pos = torch.arange(cap, device=dev)
        cell = torch.repeat_interleave(torch.arange(self.n_cells, device=dev), self._cell_capacity)
        if cell.numel() != cap:
            raise RuntimeError("CellContainer: cell capacities do not tile the storage")
        live = self._is_empty == 0
        # rank of each live slot among the live slots of its cell
        csum = torch.cumsum(live.long(), 0)
        before_cell = torch.cat([csum.new_zeros(1), csum])[self._cell_start]
        rank = csum - 1 - before_cell[cell]
        src = pos[live]
        dst = (self._cell_start[cell] + rank)[live]
        n_live = torch.zeros(self.n_cells, device=dev, dtype=torch.long).scatter_add_(
            0, cell[live], torch.ones_like(src))
        storage = torch.zeros_like(self._storage)
        storage[:, dst] = self._storage[:, src]
        a2i = torch.full_like(self._address2id, -1)
        a2i[dst] = self._address2id[src]
        is_empty = torch.ones_like(self._is_empty)
        is_empty[dst] = 0
        self._storage.copy_(storage)
        self._address2id.copy_(a2i)
        self._is_empty.copy_(is_empty)
        self._cell_size.copy_(n_live)
        self._has_holes = False
        self._packed_valid = False
        self._codes_version += 1
        self._drop_inverse_id_mapping()
        new_of_old = torch.full((cap,), -1, device=dev, dtype=torch.long)
        new_of_old[src] = dst
        return new_of_old

    def _apply(self, fn, *args, **kwargs):
        """.to() / .cuda(): the derived scan-layout copy is not a registered buffer -- drop it
        (rebuilt lazily on the new device) and follow the buffers' device"""
        out = super()._apply(fn, *args, **kwargs)
        self._arena = {}  # the buffers were re-created by fn: they own their memory again
        new_dev = str(self._storage.device)
        if new_dev != str(torch.device(self.device)) or (
                self._packed is not None and self._packed.device != self._storage.device):
            self._packed = None
            self._packed_valid = False
            self.device = new_dev
        return out

    def remove(self, ids=None, address=None):
        """Remove by id or by address.  The reference's guard `if n_removed <= self.n_items: return`
        (:381-383) is inverted, so its remove() never removes anything; this one does what the code
        after the guard intends (tombstone, -1 id, shrink the cell) and additionally moves the
        cell's last items into the holes so every cell stays dense in [start, start+size) -- the
        scan then needs no per-slot tombstone test.  Addresses of moved items change."""
        if ids is not None:
            address = self.get_address_by_id(ids)
        elif address is not None:
            address = address.to(self.device)
            assert util.check_dtype(address, torch.int64)
        else:
            raise RuntimeError("Need either ids or address")
        mask = (address >= 0) & (address < self.capacity)
        address = address[mask].unique(sorted=True)
        address = address[self._is_empty[address] == 0]
        n_removed = address.shape[0]
        if n_removed == 0:
            return
        if self._has_holes:
            # a foreign state_dict left tombstones inside some [start, start+size): the dense
            # bookkeeping below needs every cell packed first (addresses change; ids do not)
            # (remapped through the compaction's own old -> new map: going through the ids would
            # mis-resolve duplicate ids -- get_address_by_id returns one address per id -- and cost
            # O(n_ids x capacity) without the inverse table)
            address = self._compact()[address]
        cells = self.get_cell_by_address(address)
        ucells, counts = cells.unique(return_counts=True)
        old_end = (self._cell_start + self._cell_size)[ucells]
        new_end = old_end - counts
        # tail region [new_end, old_end) of every affected cell
        tail_cell = torch.repeat_interleave(torch.arange(ucells.shape[0], device=address.device), counts)
        first = torch.cumsum(counts, 0) - counts
        tail = new_end[tail_cell] + (torch.arange(n_removed, device=address.device) - first[tail_cell])
        removed_flag = torch.zeros(self.capacity, device=address.device, dtype=torch.bool)
        removed_flag[address] = True
        movers = tail[~removed_flag[tail]]                      # survivors sitting in a tail
        cell_of_adr = torch.searchsorted(ucells, cells)
        holes = address[address < new_end[cell_of_adr]]         # removed slots below the new end
        if movers.shape[0] != holes.shape[0]:
            raise RuntimeError("CellContainer.remove: cell bookkeeping is inconsistent "
                               f"({movers.shape[0]} survivors to move, {holes.shape[0]} holes)")
        if holes.shape[0]:
            self._storage[:, holes] = self._storage[:, movers]
            self._address2id[holes] = self._address2id[movers]
        self._is_empty[tail] = 1
        self._address2id[tail] = -1
        self._cell_size[ucells] -= counts
        self._packed_valid = False
        self._codes_version += 1
        self._drop_inverse_id_mapping()
        self.print_message(f"{n_removed} items has been removed", 2)
