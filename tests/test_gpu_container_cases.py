"""GPU: the cases of the reference's own container test-suite (tests/CellContainerTestCase.py:92-241
-- init, expand, add with / without ids, remove by address / by ids, add-remove interaction,
empty), written against the same public methods and imported through the `torchpq` alias, so a
TorchPQ maintainer reads the test they know.  (The reference's files cannot run as shipped --
SURVEY section 4 -- and its remove() is unreachable behind an inverted guard,
container/CellContainer.py:381-383; these are the behaviours its tests *assert*.)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CODE_SIZE, N_CELLS = 16, 64


@pytest.fixture(params=["double", "step"])
def module(request):
    import torchpq_amd.compat as compat
    compat.install_as_torchpq()
    try:
        from torchpq.container import CellContainer
        torch.manual_seed(0)
        np.random.seed(0)
        yield CellContainer(code_size=CODE_SIZE, n_cells=N_CELLS, dtype="uint8", device=DEV,
                            initial_size=32, expand_step_size=32, expand_mode=request.param,
                            use_inverse_id_mapping=True, contiguous_size=4)
    finally:
        compat.uninstall()


def make_data(n):
    return torch.randint(0, 256, (CODE_SIZE, n), device=DEV, dtype=torch.uint8)


def make_cells(module, n):
    return torch.randint(module.n_cells, (n,), device=DEV, dtype=torch.long)


def make_unique_ids(n):
    # distinct, sparse, far beyond n (NB: never np.random.choice(huge, replace=False): it
    # materialises a permutation of the whole population)
    ids = np.random.choice(1 << 22, size=n, replace=False).astype(np.int64) * 262147 + 5
    return torch.from_numpy(ids).to(DEV).long()


def test_init(module):
    assert module.capacity == N_CELLS * 32
    assert module._storage.shape == (CODE_SIZE // 4, module.capacity, 4)


def test_expand(module):
    cells = torch.from_numpy(np.random.choice(N_CELLS, 17, replace=False)).to(DEV).long()
    old_cap = module._cell_capacity[cells].clone()
    others = torch.ones(N_CELLS, dtype=torch.bool, device=DEV)
    others[cells] = False
    old_other = module._cell_capacity[others].clone()
    module.expand(cells)
    new_cap = module._cell_capacity[cells]
    if module.expand_mode == "double":
        assert torch.equal(new_cap, old_cap * 2)
    else:
        assert torch.equal(new_cap, old_cap + module.expand_step_size)
    assert torch.equal(module._cell_capacity[others], old_other)
    assert module.capacity == int(module._cell_capacity.sum())
    assert torch.equal(module._cell_start, torch.cumsum(module._cell_capacity, 0) - module._cell_capacity)


def test_add_with_ids(module):
    n = 10000
    data, cells, ids = make_data(n), make_cells(module, n), make_unique_ids(n)
    returned_ids, returned_adr = module.add(data, cells=cells, ids=ids, return_address=True)
    assert torch.equal(ids, returned_ids)
    assert torch.equal(module.get_address_by_id(ids), returned_adr)
    assert torch.equal(module.get_data_by_address(returned_adr), data)
    assert (module._is_empty[returned_adr] == 0).all()
    assert torch.equal(module.get_cell_by_address(returned_adr), cells)
    assert module.n_items == n


def test_add_without_ids(module):
    n = 10000
    data, cells = make_data(n), make_cells(module, n)
    returned_ids, returned_adr = module.add(data, cells=cells, return_address=True)
    assert torch.equal(module.get_id_by_address(returned_adr), returned_ids)
    assert torch.equal(module.get_address_by_id(returned_ids), returned_adr)
    assert torch.equal(module.get_data_by_address(returned_adr), data)
    assert (module._is_empty[returned_adr] == 0).all()
    assert torch.equal(returned_ids, torch.arange(n, device=DEV))


def test_remove_by_address(module):
    n = 10000
    data, cells, ids = make_data(n), make_cells(module, n), make_unique_ids(n)
    _, returned_adr = module.add(data, cells=cells, ids=ids, return_address=True)
    module.remove(address=returned_adr)
    assert (module.get_id_by_address(returned_adr) == -1).all()
    assert (module._is_empty[returned_adr] == 1).all()
    assert module.n_items == 0


def test_remove_by_ids(module):
    n = 10000
    data, cells, ids = make_data(n), make_cells(module, n), make_unique_ids(n)
    _, returned_adr = module.add(data, cells=cells, ids=ids, return_address=True)
    module.remove(ids=ids)
    assert (module._is_empty[returned_adr] == 1).all()
    assert (module.get_address_by_id(ids) == -1).all()
    assert (module.get_id_by_address(returned_adr) == -1).all()


def test_add_remove_interaction(module):
    n = 1000
    data, cells, ids = make_data(n), make_cells(module, n), make_unique_ids(n)
    _, returned_adr = module.add(data, cells=cells, ids=ids, return_address=True)
    remove_idx = torch.from_numpy(np.random.choice(n, n // 2, replace=False)).to(DEV)
    keep = torch.ones(n, dtype=torch.bool, device=DEV)
    keep[remove_idx] = False
    module.remove(address=returned_adr[remove_idx])
    assert module.n_items == n - n // 2
    # the survivors are still found by id, with their codes
    adr = module.get_address_by_id(ids[keep])
    assert (adr >= 0).all() and torch.equal(module.get_data_by_address(adr), data[:, keep])
    assert (module.get_address_by_id(ids[~keep]) == -1).all()
    new_data, new_cells = make_data(n // 2), make_cells(module, n // 2)
    new_ids, new_adr = module.add(new_data, cells=new_cells, return_address=True)
    assert torch.equal(module.get_data_by_address(new_adr), new_data)
    assert torch.equal(module.get_id_by_address(new_adr), new_ids)
    assert torch.equal(module.get_address_by_id(new_ids), new_adr)
    assert module.n_items == n


def test_empty(module):
    assert module.n_items == 0
    assert (module._cell_size == 0).all()
    module.add(make_data(100), cells=make_cells(module, 100))
    module.empty()
    assert module.n_items == 0 and (module._cell_size == 0).all() and (module._is_empty == 1).all()


def test_linear_get_address_by_id_without_inverse_mapping():
    """use_inverse_id_mapping=False: ids are resolved by the linear search of
    the reference (kernels/cuda/get_address_by_id.cu:8-44, tpq_get_address_by_id) -- no table is
    built -- and agree with the table path; remove(ids=...) works through it."""
    from torchpq_amd.container import CellContainer
    torch.manual_seed(3)
    kw = dict(code_size=CODE_SIZE, n_cells=N_CELLS, dtype="uint8", device=DEV, initial_size=16,
              expand_step_size=16, expand_mode="double", contiguous_size=4)
    lin = CellContainer(use_inverse_id_mapping=False, **kw)
    tab = CellContainer(use_inverse_id_mapping=True, **kw)
    n = 5000
    data = make_data(n)
    cells = torch.randint(N_CELLS, (n,), device=DEV, dtype=torch.long)
    ids = torch.randperm(10 ** 6, device=DEV)[:n] * 7 + 3
    for c in (lin, tab):
        c.add(data[:, :3000].contiguous(), cells[:3000], ids[:3000])
        c.add(data[:, 3000:].contiguous(), cells[3000:], ids[3000:])
    probe = torch.cat([ids[torch.randperm(n, device=DEV)[:700]],
                       torch.tensor([-5, -1, 0, 1, 2, 10 ** 9], device=DEV)])
    a_lin, a_tab = lin.get_address_by_id(probe), tab.get_address_by_id(probe)
    assert lin._id2address is None and lin._sparse_id_map is None     # no table was created
    assert torch.equal(a_lin, a_tab)
    assert bool((a_lin[-6:] == -1).all()) and bool((a_lin[:700] >= 0).all())
    assert torch.equal(lin._address2id[a_lin[:700]], probe[:700])
    assert torch.equal(lin.get_address_by_id(probe.reshape(2, -1)), a_lin.reshape(2, -1))
    gone = ids[::5].contiguous()
    lin.remove(ids=gone)
    tab.remove(ids=gone)
    assert lin.n_items == tab.n_items == n - gone.shape[0]
    assert torch.equal(lin._address2id, tab._address2id) and torch.equal(lin._storage, tab._storage)
    assert bool((lin.get_address_by_id(gone) == -1).all())
    assert lin.get_address_by_id(torch.empty(0, dtype=torch.long, device=DEV)).shape == (0,)


def test_base_container_expand_grows_the_id_table():
    """BaseContainer.expand (BaseContainer.py:112-127): one step of free addresses, the step doubling
    first in "double" mode; ids already stored keep their addresses."""
    from torchpq_amd.container.BaseContainer import BaseContainer

    class Bare(BaseContainer):
        def add(self):
            pass

        def remove(self):
            pass

    for mode, want in (("double", [16, 16 + 16, 16 + 16 + 32]), ("step", [16, 24, 32])):
        c = Bare(device=DEV, initial_size=16, expand_step_size=8, expand_mode=mode)
        c._address2id[:3] = torch.tensor([7, 5, 9], device=DEV)
        sizes = [c.capacity]
        for _ in range(2):
            c.expand()
            sizes.append(c.capacity)
        assert sizes == want, (mode, sizes)
        assert c._address2id[:3].tolist() == [7, 5, 9] and bool((c._address2id[3:] == -1).all())
        assert c.get_id_by_address(torch.tensor([1, sizes[-1] - 1], device=DEV)).tolist() == [5, -1]


def test_growth_arenas_keep_the_layout_and_exact_state_dict():
    """Large containers grow inside two geometrically sized arenas that swap roles (CellContainer._grow):
    the layout after a sequence of adds equals the one a fresh-allocation container produces, buffers
    are contiguous prefixes of the arenas, few allocations happen, and state_dict() hands out
    exact-size tensors."""
    from torchpq_amd.container import CellContainer

    def build(arena_min_bytes):
        torch.manual_seed(3)
        c = CellContainer(code_size=CODE_SIZE, n_cells=N_CELLS, dtype="uint8", device=DEV, initial_size=8,
                          expand_step_size=8, expand_mode="step", use_inverse_id_mapping=True,
                          contiguous_size=4)  # "step": every add below grows the storage by ~2 %
        c.arena_min_bytes = arena_min_bytes
        allocs, grows = set(), 0
        for step in range(40):  # one large add, then many small ones: frequent growth by a few %
            n = 60000 if step == 0 else 1500
            cap = c.capacity
            c.add(torch.randint(0, 256, (CODE_SIZE, n), device=DEV, dtype=torch.uint8),
                  torch.randint(N_CELLS, (n,), device=DEV))
            grows += int(c.capacity != cap)
            allocs.add(c._storage.untyped_storage().data_ptr())
        return c, (allocs, grows)

    plain, plain_allocs = build(1 << 60)
    arena, arena_allocs = build(0)
    for name in ("_storage", "_address2id", "_is_empty", "_cell_start", "_cell_size", "_cell_capacity"):
        assert torch.equal(getattr(plain, name), getattr(arena, name)), name
    assert arena._storage.is_contiguous() and arena._storage.shape == plain._storage.shape
    assert arena._storage.untyped_storage().nbytes() >= arena._storage.numel()
    ptrs, grows = arena_allocs
    assert grows >= 10 and len(ptrs) <= grows // 2, (grows, len(ptrs))  # growth re-uses the two arenas
    sd = arena.state_dict()
    for name in ("_storage", "_address2id", "_is_empty"):
        t = sd[name]
        assert t.untyped_storage().nbytes() == t.numel() * t.element_size(), name
        assert torch.equal(t, getattr(plain, name))
    # the spare side can be dropped and growth still works
    arena.release_spare()
    n = 40000
    data = torch.randint(0, 256, (CODE_SIZE, n), device=DEV, dtype=torch.uint8)
    cells = torch.randint(N_CELLS, (n,), device=DEV)
    arena.add(data, cells)
    plain.add(data, cells)
    assert torch.equal(plain._storage, arena._storage) and torch.equal(plain._address2id, arena._address2id)
    # load_state_dict detaches the container from its arenas
    arena.load_state_dict(plain.state_dict())
    assert arena._arena == {}
