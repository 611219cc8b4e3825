"""GPU: SURVEY 8 row f-2 at size -- a 100 M-vector index built through IVFPQIndex.add in chunks;
placement vs the sort-by-cell oracle on sampled cells, stored codes vs encode() on sampled chunks
(tools/build_100m.py).  A 3 M-vector version of the same check runs everywhere."""
import importlib.util
import os

import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _tool():
    spec = importlib.util.spec_from_file_location("build_100m", os.path.join(ROOT, "tools", "build_100m.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bulk_add_3m_placement_and_codes():
    tool = _tool()
    n, chunk = 3_000_000, 1 << 19
    idx, cells_all, centers, t = tool.build(n, chunk, n_cells=2048, n_train=200_000)
    assert idx.n_items == n and idx.max_id == n - 1
    assert t["chunks_that_grew_the_storage"] >= 2          # default initial_size: cells double
    assert tool.check_placement(idx, cells_all, n_sample_cells=64) > 0
    assert tool.check_codes(idx, centers, chunk, chunk_ids=(0, 3, 5), n_total=n) > 0


def test_bulk_add_100m_placement_and_codes():
    free, _ = torch.cuda.mem_get_info()
    if free < 60 * 2 ** 30:
        pytest.skip("needs ~30 GB of HBM")
    tool = _tool()
    n, chunk = 100_000_000, 1 << 20
    idx, cells_all, centers, t = tool.build(n, chunk)
    assert idx.n_items == n and idx.max_id == n - 1 and idx.capacity >= n
    assert tool.check_placement(idx, cells_all, n_sample_cells=48) > 100_000
    assert tool.check_codes(idx, centers, chunk, chunk_ids=(0, 37, 95), n_total=n) == 2 * chunk + (n - 95 * chunk)
    # addresses beyond the reference kernel's fp32-exact range are in use
    assert int(idx._cell_start[-1]) > 2 ** 24
