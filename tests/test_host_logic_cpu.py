"""CPU: host-side policy code that needs no GPU (split heuristics, layout policy, alias table)."""
import pytest


def test_scan_split_heuristic():
    from torchpq_amd.kernels import IVFPQTopkHip
    scan = IVFPQTopkHip(m=64)
    scan.n_cus = 256                       # MI355X; avoids the device query
    assert scan._n_split(10000, "cuda:0") == 1            # large batches: one workgroup per query
    assert scan._n_split(512, "cuda:0") == 1
    assert scan._n_split(256, "cuda:0") == 2              # fill 2 workgroups per CU
    assert scan._n_split(1, "cuda:0") == 64               # capped
    # with a work hint every wave keeps >= 4 tiles: 32 cells x 1024 slots = 512 tiles -> 16 splits
    assert scan._n_split(1, "cuda:0", slots_hint=32 * 1024) == 16
    assert scan._n_split(1, "cuda:0", slots_hint=100) == 1
    big = IVFPQTopkHip(m=120)              # one 16-wave workgroup per CU
    big.n_cus = 256
    assert big._n_split(1000, "cuda:0") == 1 and big._n_split(64, "cuda:0") == 4
    assert big._n_split(1, "cuda:0", slots_hint=64 * 1024) == 16


def test_scan_layout_policy_and_instantiated_m():
    from torchpq_amd.index import IVFPQIndex
    from torchpq_amd.kernels import PACKED_M, packed_chunk_width
    assert IVFPQIndex.packed_min_subvectors == 56 and IVFPQIndex.packed_max_short_subvectors == 24
    assert all(m % 4 == 0 for m in PACKED_M) and 64 in PACKED_M and 120 in PACKED_M
    assert [packed_chunk_width(m) for m in (4, 8, 12, 16, 24, 120, 128)] == [4, 8, 4, 16, 8, 8, 16]
    # the list in the Python layer is the one compiled into the library (scan_device.h)
    import os
    import re
    from conftest import ROOT
    text = open(os.path.join(ROOT, "torchpq_amd", "csrc", "scan_device.h")).read()
    m_list = re.search(r"#define TPQ_PACKED_M_LIST\(X\) \\\n(.*)\n", text).group(1)
    assert tuple(int(x) for x in re.findall(r"X\((\d+)\)", m_list)) == PACKED_M
    build = open(os.path.join(ROOT, "torchpq_amd", "csrc", "build.sh")).read()
    loop = re.search(r"for m in ([\d ]+);", build).group(1)
    assert tuple(int(x) for x in loop.split()) == PACKED_M


def test_alias_table_points_at_existing_wrappers():
    import torchpq_amd.kernels as K
    from torchpq_amd.compat import KERNEL_ALIASES, SUBMODULES
    for ref_name, hip_name in KERNEL_ALIASES.items():
        assert ref_name.endswith("Cuda") and hasattr(K, hip_name), (ref_name, hip_name)
    import importlib
    for name in SUBMODULES:
        importlib.import_module("torchpq_amd." + name)


def test_cpu_devices_are_refused():
    from torchpq_amd.index import FlatIndex, IVFPQIndex
    for ctor in (lambda: IVFPQIndex(d_vector=32, n_subvectors=8, device="cpu"),
                 lambda: FlatIndex(d_vector=32, device="cpu")):
        with pytest.raises((RuntimeError, AssertionError)):
            ctor()
