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
    short = IVFPQTopkHip(m=16)             # four 4-wave workgroups per CU (r02)
    short.n_cus = 256
    assert short._n_split(1024, "cuda:0") == 1 and short._n_split(512, "cuda:0") == 2
    assert short._n_split(1, "cuda:0", slots_hint=32 * 1024) == 32   # 512 tiles / (4 waves x 4 tiles)
    big = IVFPQTopkHip(m=120)              # one 16-wave workgroup per CU
    big.n_cus = 256
    assert big._n_split(1000, "cuda:0") == 1 and big._n_split(64, "cuda:0") == 4
    assert big._n_split(1, "cuda:0", slots_hint=64 * 1024) == 16


def test_scan_layout_policy_and_instantiated_m():
    from torchpq_amd.index import IVFPQIndex
    from torchpq_amd.kernels import PACKED_M, packed_chunk_width
    # r02: the scan layout is kept at every instantiated m (it wins at 28..48 too since the
    # slots-per-lane policy was extended)
    assert IVFPQIndex.packed_min_subvectors == 0 and IVFPQIndex.packed_max_short_subvectors == 24
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
    assert tuple(sorted(int(x) for x in loop.split())) == PACKED_M   # (build.sh compiles the longest units first)


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


def test_texmex_readers_round_trip(tmp_path):
    """fvecs / ivecs (the SIFT1M / GIST1M file format bench.py --data-dir reads)"""
    import numpy as np
    from torchpq_amd import datasets
    rng = np.random.default_rng(0)
    x = rng.standard_normal((37, 128)).astype(np.float32)
    gt = rng.integers(0, 1000, (37, 100)).astype(np.int32)
    d = tmp_path / "sift"
    d.mkdir()
    datasets.write_fvecs(d / "sift_base.fvecs", x)
    datasets.write_fvecs(d / "sift_query.fvecs", x[:5])
    datasets.write_ivecs(d / "sift_groundtruth.ivecs", gt)
    assert np.array_equal(datasets.read_fvecs(d / "sift_base.fvecs"), x)
    assert np.array_equal(datasets.read_fvecs(d / "sift_base.fvecs", 10), x[:10])
    assert np.array_equal(datasets.read_ivecs(d / "sift_groundtruth.ivecs"), gt)
    raw = np.fromfile(d / "sift_base.fvecs", dtype="<i4")
    assert raw[0] == 128 and raw[129] == 128          # int32 dimension in front of every vector
    p = datasets.find_texmex(str(tmp_path), "sift")
    assert p["base"].endswith("sift_base.fvecs") and p["learn"] is None and p["groundtruth"]
    assert datasets.find_texmex(str(tmp_path), "gist") is None
    with open(d / "bad.fvecs", "wb") as f:
        f.write(b"\\x80\\x00\\x00\\x00" + b"\\x00" * 7)
    import pytest
    with pytest.raises(ValueError):
        datasets.read_fvecs(d / "bad.fvecs")


def test_bench_launch_command_and_defaults():
    """bench.py --gpus N (N > 1, no launcher) re-executes itself as one rank per GPU"""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cmd = bench.launch_command(4, ["--gpus", "4", "--steps", "5"], port=29512)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd and "29512" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "5"] and cmd[-5].endswith("bench.py")
    a = bench.parse_args([])
    assert (a.gpus, a.workload, a.n_base, a.n_cells, a.n_probe, a.m, a.k) == (1, "c2", 1_000_000, 1024, 32, 64, 100)
    a = bench.parse_args(["--workload", "c4", "--gpus", "8"])
    assert (a.n_base, a.n_cells, a.n_probe) == (100_000_000, 16384, 64)
    assert len(bench.source_fingerprint()) == 16
    assert a.scaling == "weak" and bench.parse_args(["--scaling", "strong"]).scaling == "strong"


def test_bench_sets_the_rccl_environment_on_both_entry_paths():
    """HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC, which RCCL needs on this driver) must
    be in place before the first HIP call under the DRIVER's `torch.distributed.run ... bench.py`
    (WORLD_SIZE already set: main() never goes through self_launch) as well as under self_launch."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    probe = (
        "import os, sys, importlib.util\n"
        "os.environ.pop('HSA_ENABLE_IPC_MODE_LEGACY', None)\n"
        "spec = importlib.util.spec_from_file_location('bench_mod', sys.argv[1])\n"
        "assert 'torch' not in sys.modules\n"
        "b = importlib.util.module_from_spec(spec)\n"
        "import builtins\n"
        "orig = builtins.__import__\n"
        "seen = {}\n"
        "def hook(name, *a, **k):\n"
        "    if name == 'torch' and 'at_torch_import' not in seen:\n"
        "        seen['at_torch_import'] = os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY')\n"
        "    return orig(name, *a, **k)\n"
        "builtins.__import__ = hook\n"
        "spec.loader.exec_module(b)\n"
        "builtins.__import__ = orig\n"
        "print('AT_IMPORT', seen.get('at_torch_import'))\n"
        "print('AFTER', os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY'), os.environ.get('MASTER_ADDR'))\n"
        "env = b.rccl_env({})\n"
        "print('CHILD', env['HSA_ENABLE_IPC_MODE_LEGACY'], env['MASTER_ADDR'])\n")
    # the driver's entry path: a rank of torch.distributed.run (WORLD_SIZE set), env var absent
    env = dict(os.environ, WORLD_SIZE="8", RANK="3", LOCAL_RANK="3")
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    env.pop("MASTER_ADDR", None)
    out = subprocess.run([sys.executable, "-c", probe, os.path.join(ROOT, "bench.py")], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-1500:]
    assert "AT_IMPORT 0" in out.stdout, out.stdout      # set before torch (hence HIP) is imported
    assert "AFTER 0 127.0.0.1" in out.stdout, out.stdout
    assert "CHILD 0 127.0.0.1" in out.stdout, out.stdout  # what self_launch hands its ranks
    # an explicit setting by the operator is respected
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "1"
    out = subprocess.run([sys.executable, "-c", probe.replace(
        "os.environ.pop('HSA_ENABLE_IPC_MODE_LEGACY', None)\n", ""), os.path.join(ROOT, "bench.py")],
        env=env, capture_output=True, text=True, timeout=300)
    assert "AFTER 1" in out.stdout, out.stdout + out.stderr[-500:]


def test_assign_path_policy_and_host_side_shape_functions():
    """which kernel the Lloyd loop's assign runs on (MultiKMeans._assign_path) and the pure host
    functions behind it (no GPU needed: *_supported / *_workspace_bytes only inspect shapes)"""
    from torchpq_amd import _lib
    from torchpq_amd.clustering import MultiKMeans
    lib = _lib.load()
    mk = MultiKMeans(n_clusters=256)
    # codebook-sized problems inside fit(): exact-label selection kernel; outside fit(): fp32
    assert mk._assign_path(64, 64, 1_000_000, 256, training=True) == "select"
    assert mk._assign_path(64, 64, 1_000_000, 256, training=False) == "fp32"
    assert mk._assign_path(64, 2, 100_000, 256, training=True) == "fp32"       # SIFT's d_sub = 2
    assert mk._assign_path(120, 8, 100_000, 256, training=True) == "fp32"      # GIST's d_sub = 8
    # one problem, many centroids: coarse assign; batched with > 256 centroids: split kernel
    assert mk._assign_path(1, 128, 1_000_000, 16384, training=True) == "coarse"
    assert mk._assign_path(1, 960, 500_000, 1024, training=True) == "coarse"   # wide vectors: the GEMM-shaped cascade
    assert mk._assign_path(1, 1025, 500_000, 1024, training=True) == "fp32"    # d > 1024
    assert mk._assign_path(4, 64, 100_000, 512, training=True) == "bf16x3"
    assert MultiKMeans(n_clusters=256, assign_precision="fp32")._assign_path(64, 64, 10 ** 6, 256, True) == "fp32"
    # shape functions
    assert lib.tpq_max_sim_select_supported(64, 64, 1_000_000, 256) == 1
    assert lib.tpq_max_sim_select_supported(64, 65, 1000, 256) == 0
    assert lib.tpq_max_sim_select_supported(64, 64, 1000, 257) == 0
    ws = lib.tpq_max_sim_select_workspace_bytes(64, 64, 1_000_000, 256)
    assert 64 * 1_000_000 * 4 <= ws <= 64 * 1_000_000 * 4 + (8 << 20)         # the lists dominate
    assert lib.tpq_coarse_assign_supported(128, 1 << 20, 16384) == 1
    assert lib.tpq_coarse_assign_supported(129, 1000, 16) == 1 and lib.tpq_coarse_assign_supported(960, 10 ** 6, 16384) == 1
    assert lib.tpq_coarse_assign_supported(1025, 1000, 16) == 0 and lib.tpq_coarse_assign_supported(960, 1 << 28, 16) == 0
    ws960 = lib.tpq_coarse_assign_workspace_bytes(960, 10 ** 6, 16384)
    assert 2 * 960 * 10 ** 6 <= ws960 <= 4 * 4 * 960 * 10 ** 6                # fp16 pieces + the listed points' copies
    assert lib.tpq_coarse_assign_supported(128, 1 << 23, 16) == 0              # padded slice >= 2 GiB
    ws = lib.tpq_coarse_assign_workspace_bytes(128, 1 << 20, 16384)
    off = lib.tpq_coarse_assign_count_offset(128, 1 << 20, 16384)
    frag_bytes = (16384 // 128) * 4 * 17 * 1024                                # 128 chunks x 4 units x 17 KiB
    assert off == frag_bytes + 256 and ws >= frag_bytes + (1 << 20) * 4
    assert lib.tpq_max_sim_split_supported(64, 1_000_000, 256) == 1 and lib.tpq_max_sim_split_supported(65, 10, 4) == 0


def test_scan_route_rules():
    """tpq_ivfpq_scan_route: the library's own routing rule, host only (nothing is launched) -- which kernels a scan call
    runs (reference dispatch by k and m: fn/IVFPQTopk.py:39-104, kernels/IVFPQTopkCuda.py:81-142)"""
    from torchpq_amd import kernels as K

    def r(m, nq, k, ds=2, n_split=1, n_probe=32, hint=None, has_lut=False, packed=True, residual=False, tickets=None):
        return K.IVFPQTopkHip(m=m).route(nq, k, n_split, ds, n_probe, hint, has_lut, packed, tickets, residual)

    # the headline and the 100 M-slot workload: four-wave workgroups over the 16-bit table
    assert r(64, 10000, 100, hint=32 * 977) == "dump_sel16"
    assert r(64, 10000, 100, n_probe=64, hint=64 * 6103) == "dump_sel16"
    assert r(64, 10000, 300, hint=32 * 977) == "dump_sel16_w8"      # k in (248, 504] on long cells: eight waves
    assert r(64, 10000, 500, hint=32 * 977) == "dump_sel16_w8"      # round 6: the finish kernel's exact list of 1 024
    assert r(64, 10000, 600) == "pools"
    assert r(64, 1023, 100) == "one_launch_finish"                  # below the route's batch size
    assert r(64, 10000, 100, has_lut=True) == "one_launch_finish"   # m = 64 needs query + codebook for the finish
    assert r(64, 10000, 100, ds=4) == "one_launch_finish"           # m * ds > 128
    assert r(64, 16, 100, n_split=8, tickets=False) == "sorted_lists" and r(64, 16, 100, n_split=8) == "one_launch_finish"
    # round 6: the short codes
    for m, ds in ((32, 4), (16, 8), (8, 16), (32, 1)):
        assert r(m, 10000, 100, ds=ds, hint=32 * 244) == "dump_f32"
        # the caller's table: entries gathered per survivor -- behind long scans only
        assert r(m, 10000, 100, has_lut=True, hint=32 * 977) == "dump_f32"
        assert r(m, 10000, 100, has_lut=True, hint=32 * 244) == "one_launch_finish"
        assert r(m, 10000, 100, has_lut=True) == "one_launch_finish"
        assert r(m, 1000, 100, ds=ds) == "one_launch_finish"
    assert r(32, 10000, 100, ds=8) == "one_launch_finish"           # fused, m * ds > 128: the codebook would not fit
    assert r(32, 10000, 300, ds=4) == "pools"                       # k > 248 at m <= 32
    assert r(24, 10000, 100, ds=4) == "one_launch_finish"           # block structure 16 + 8: not built for the route
    assert r(120, 1000, 100, has_lut=True, n_probe=64) == "one_launch_finish"
    assert r(64, 10000, 100, residual=True) == "sorted_lists"
    assert r(64, 10000, 100, packed=False) == "reference_layout"
    assert r(64, 10000, 1020, has_lut=True) == "reference_layout"   # no room for the candidate band next to k
    assert r(64, 0, 100) == "rejected"
