"""CPU: the C-ABI library builds, loads, and exports every symbol include/torchpq_amd.h declares
(no compute calls -- there is no GPU here)."""
import os
import re

import pytest

from conftest import ROOT


def test_library_loads_and_exports_header_symbols():
    from torchpq_amd import _lib
    lib = _lib.load()
    assert lib.tpq_version() == 200
    header = open(os.path.join(ROOT, "include", "torchpq_amd.h")).read()
    declared = set(re.findall(r"\b(tpq_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name


def test_argument_validation_without_a_gpu():
    """Validation happens before any HIP call, so it can be exercised on the CPU box."""
    from torchpq_amd import _lib
    lib = _lib.load()
    rc = lib.tpq_topk_select(None, None, None, 1, 10, 1, None)
    assert rc == -1 and "null pointer" in _lib.last_error()
    rc = lib.tpq_ivfpq_pack_codes(None, None, 10, 8, 0, 10, None)
    assert rc == -1
    # flags + error bounds + per-wave candidate lists (8 waves per workgroup at m <= 64, 16 above)
    assert lib.tpq_ivfpq_scan_workspace_bytes(100, 100, 1, 64) == 2 * 512 + 100 * 8 * 128 * 8
    assert lib.tpq_ivfpq_scan_workspace_bytes(100, 100, 4, 120) == 2 * 512 + 100 * 4 * 16 * 128 * 8
    assert lib.tpq_compute_centroids_workspace_bytes(2, 3, 5) == (2 * 3 * 5 + 2 * 5) * 4


def test_product_refuses_cpu_tensors_and_has_no_oracle_import():
    import pytest
    import torch
    from torchpq_amd import kernels
    from torchpq_amd._lib import TorchPQAmdError
    with pytest.raises(TorchPQAmdError):
        kernels.TopkSelectHip()(torch.zeros(2, 8), k=1)
    with pytest.raises(RuntimeError):
        from torchpq_amd.index import IVFPQIndex
        IVFPQIndex(32, 8, 16, device="cpu")
    # the oracle is test infrastructure: nothing in the package may reference it
    pkg = os.path.join(ROOT, "torchpq_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
                assert "/root/reference" not in src, f


def test_torchpq_import_alias():
    """torchpq_amd.compat.install_as_torchpq(): reference import paths and wrapper names resolve to
    this package (CPU: names only)."""
    import sys
    import torchpq_amd.compat as compat
    # (the live-pin test may have imported the real reference into this process: set it aside)
    parked = {k: sys.modules.pop(k) for k in list(sys.modules) if k == "torchpq" or k.startswith("torchpq.")}
    compat.install_as_torchpq()
    try:
        from torchpq.index import IVFPQIndex
        from torchpq.kernels import (ComputeCentroidsCuda, GetDivByAddressV2Cuda, GetIOACuda,
                                     GetWriteAddressV2Cuda, IVFPQTop1Cuda, IVFPQTopkCuda, MaxSimCuda,
                                     PQDecodeCuda, Top1SelectCuda, Top32SelectCuda, TopkSelectCuda)
        import torchpq
        from torchpq_amd import kernels as K
        assert IVFPQIndex.__module__ == "torchpq_amd.index.IVFPQIndex"
        assert IVFPQTopkCuda is K.IVFPQTopkHip and MaxSimCuda is K.MaxSimHip
        assert torchpq.codec.PQCodec.__module__.startswith("torchpq_amd")
        assert callable(torchpq.metric.negative_squared_l2_distance)
        with pytest.raises(RuntimeError):
            sys.modules["torchpq"].__torchpq_amd_alias__ = False
            compat.install_as_torchpq()          # would shadow a "foreign" torchpq
        sys.modules["torchpq"].__torchpq_amd_alias__ = True
    finally:
        compat.uninstall()
    assert "torchpq" not in sys.modules and "torchpq.index" not in sys.modules
    sys.modules.update(parked)
