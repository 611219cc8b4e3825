"""GPU: the replica path of SURVEY 8e on one device -- two ranks share cuda:0 and rendezvous over
gloo (RCCL refuses two ranks on one GPU): rank 0 builds an index, replicate_index() broadcasts
it, the non-source rank load_state_dict()s it and its search() must equal rank 0's bit for bit.
Also: bench.py --gpus 2 launches its own ranks (TPQ_BENCH_ONE_DEVICE validation hook)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from torchpq_amd import distributed as tpd
    from torchpq_amd.index import IVFPQIndex
    dev = "cuda:0"
    d, m, n_cells, n, nq, k = 64, 16, 32, 20000, 300, 20
    rng = np.random.default_rng(3)
    base = np.abs(rng.standard_normal((d, n)) * 25).astype(np.float32)
    queries = torch.from_numpy(np.abs(rng.standard_normal((d, nq)) * 25).astype(np.float32)).to(dev)
    idx = IVFPQIndex(d_vector=d, n_subvectors=m, n_cells=n_cells, initial_size=8, device=dev)
    if rank == 0:
        np.random.seed(3)
        xb = torch.from_numpy(base).to(dev)
        idx.train(xb)
        idx.add(xb[:, :12000].contiguous(), ids=torch.arange(12000, device=dev) * 2 + 5)
        idx.add(xb[:, 12000:].contiguous(), ids=torch.arange(12000, n, device=dev) * 2 + 5)
        idx.remove(ids=torch.arange(100, device=dev) * 2 + 5)
        idx.n_probe = 6
        idx.use_smart_probing = True
        idx.smart_probing_temperature = 20.0
    else:
        assert not idx.vq_codec.is_trained
    tpd.replicate_index(idx, src=0)
    assert idx.vq_codec.is_trained and idx.pq_codec.is_trained
    assert idx.n_probe == 6 and idx.use_smart_probing and idx.smart_probing_temperature == 20.0
    assert idx.n_items == n - 100 and idx.max_id == (n - 1) * 2 + 5
    # every rank searches the FULL batch here (so results are comparable) and its own shard
    v, i = idx.search(queries, k=k)
    vs, is_ = tpd.sharded_search(lambda x, kk: idx.search(x, k=kk), queries, k, gather=True)
    torch.cuda.synchronize()
    pair = [None, None]
    dist.all_gather_object(pair, (v.cpu(), i.cpu(), vs.cpu(), is_.cpu()))
    if rank == 0:
        (v0, i0, g0, gi0), (v1, i1, g1, gi1) = pair
        ret["replica_equal"] = bool(torch.equal(v0, v1) and torch.equal(i0, i1))
        ret["gather_equal"] = bool(torch.equal(g0, v0) and torch.equal(gi0, i0)
                                   and torch.equal(g1, v0) and torch.equal(gi1, i0))
        ret["finite"] = bool(torch.isfinite(v0[:, 0]).all()) and int(i0.min()) >= 5
    dist.destroy_process_group()


def test_replicate_index_two_ranks_one_gpu():
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29600 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert ret.get("replica_equal") is True
    assert ret.get("gather_equal") is True
    assert ret.get("finite") is True


def test_bench_gpus2_self_launches_under_the_one_device_hook():
    """python bench.py --gpus 2 with no launcher around it: spawns 2 ranks itself and rank 0 prints
    one JSON line with n_gpus=2 (reduced sizes; the hook puts both ranks on cuda:0 over gloo)"""
    env = dict(os.environ, TPQ_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
         "--nq", "512", "--n-base", "60000", "--n-train", "20000", "--n-cells", "64", "--n-probe", "8"],
        env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["world_size_seen"] == 2
    assert rec["config"]["collective_backend"] == "gloo"
    assert rec["value"] > 0 and rec["scaling"] == "weak" and rec["steps"] == 2
    assert "cpu_baseline" not in rec and "secondary" not in rec
    # weak: 512 queries per rank; the strong-scaling leg (ONE 512-query batch split 256 + 256) rides
    # on the same line, with the per-rank spread and the broadcast volume
    cfg = rec["config"]
    assert cfg["n_query_per_rank"] == 512 and cfg["index_broadcast_bytes"] > 60000 * 64
    assert cfg["index_broadcast_GBps"] > 0 and cfg["collective_fallback_reason"] is None
    assert 0 < cfg["ms_per_step_rank_min"] <= cfg["ms_per_step_rank_max"] == rec["ms_per_step"]
    other = rec["other_scaling"]
    assert other["scaling"] == "strong" and other["queries_per_step_all_ranks"] == 512 and other["value"] > 0
    # VERDICT r5 #1c: the N-GPU line carries its own parity figures -- rank 0's timed results against the C oracle on
    # a 256-query sample, and a digest per rank of its shard of the shared batch against rank 0's unsharded search
    oc = rec["oracle_check"]
    assert oc["queries_checked"] == 256 and oc["ids_equal_to_oracle"] == 1.0 and oc["values_bit_equal"] is True
    assert rec["recall_vs_ref@100"] == 1.0
    rc = rec["replica_check"]
    assert len(rc["shard_digests"]) == 2 and rc["all_ranks_bit_equal_to_rank0_unsharded"] is True
    assert rc["shard_digests"][0] != rc["shard_digests"][1]          # (different shards: different rows)


def test_bench_gpus2_strong_scaling_is_the_headline_when_asked():
    env = dict(os.environ, TPQ_BENCH_ONE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
         "--scaling", "strong", "--nq", "511", "--n-base", "60000", "--n-train", "20000", "--n-cells", "64",
         "--n-probe", "8"],
        env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["scaling"] == "strong" and rec["n_gpus"] == 2
    assert rec["config"]["n_query_per_rank"] == 256          # rank 0's share of 511 (256 + 255)
    assert rec["other_scaling"]["scaling"] == "weak"
    assert rec["other_scaling"]["queries_per_step_all_ranks"] == 2 * 511
    assert abs(rec["value"] - 511 * 2 / (rec["ms_per_step"] * 2e-3)) / rec["value"] < 0.01


def _bench2(extra_env, *extra_args, timeout=600):
    env = dict(os.environ, TPQ_BENCH_ONE_DEVICE="1", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run(
        [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
         "--nq", "512", "--n-base", "60000", "--n-train", "20000", "--n-cells", "64", "--n-probe", "8",
         *extra_args], env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_rccl_probe_failure_falls_back_to_gloo_and_says_so():
    """the first 8-GPU run happens unattended: a broken RCCL transport must end in a measured line over the
    gloo-staged broadcast with the reason on the record -- not in a hang (searching needs no collective)"""
    out = _bench2({"TPQ_BENCH_FAIL_RCCL": "1"})
    assert out.returncode == 0, out.stderr[-2000:]
    rec = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 2 and rec["value"] > 0
    assert rec["config"]["collective_backend"] == "gloo"
    assert "TPQ_BENCH_FAIL_RCCL" in rec["config"]["collective_fallback_reason"] \
        or "another rank" in rec["config"]["collective_fallback_reason"]


def test_bench_deadline_prints_one_error_line_instead_of_hanging():
    out = _bench2({}, "--deadline", "0.05")
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode != 0
    assert len(lines) == 1, out.stdout[-2000:]
    rec = json.loads(lines[0])
    assert rec["value"] is None and "deadline" in rec["error"] and rec["world_size_seen"] == 2


_RCCL_ONE = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import numpy as np
import torch
import torch.distributed as dist
from torchpq_amd import distributed as tpd
from torchpq_amd.index import IVFPQIndex
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
g = tpd.init_groups(dev, want_rccl=True, timeout_s=120)
out = {"bulk_backend": g.bulk_backend, "bulk_error": g.bulk_error}
rng = np.random.default_rng(5)
xb = torch.from_numpy(np.abs(rng.standard_normal((32, 6000)) * 25).astype(np.float32)).to(dev)
np.random.seed(5)
idx = IVFPQIndex(d_vector=32, n_subvectors=8, n_cells=16, initial_size=8, device="cuda:0")
idx.train(xb)
idx.add(xb)
idx.n_probe = 4
v0, i0 = idx.search(xb[:, :64].contiguous(), k=5)
torch.cuda.synchronize()
tpd.host_barrier()
tpd.replicate_index(idx, src=0, bulk_group=g.bulk, chunk_bytes=1 << 16)   # (many chunks: the chunked path on RCCL)
torch.cuda.synchronize()
v1, i1 = idx.search(xb[:, :64].contiguous(), k=5)
big = torch.arange(1 << 22, device=dev, dtype=torch.int32)                  # 16 MiB through one RCCL broadcast
if g.bulk is not None:
    dist.broadcast(big, src=0, group=g.bulk)
    t = torch.ones(4, device=dev)
    dist.all_reduce(t, group=g.bulk)
    out["all_reduce"] = t.tolist()
torch.cuda.synchronize()
out["replicated_bytes"] = int(idx.replicated_bytes)
out["same"] = bool(torch.equal(v0, v1) and torch.equal(i0, i1) and int(big[-1]) == (1 << 22) - 1)
print(json.dumps(out))
dist.destroy_process_group()
'''


_NCCL_ONLY = r'''
import json, os, sys
sys.path.insert(0, sys.argv[1])
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[2], RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import numpy as np
import torch
import torch.distributed as dist
from torchpq_amd import distributed as tpd
from torchpq_amd.index import IVFPQIndex
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)          # what INTEGRATION.md's shortest recipe does: NO CPU backend
out = {"object_device": str(tpd.object_device(None))}
rng = np.random.default_rng(6)
xb = torch.from_numpy(np.abs(rng.standard_normal((32, 6000)) * 25).astype(np.float32)).to(dev)
np.random.seed(6)
idx = IVFPQIndex(d_vector=32, n_subvectors=8, n_cells=16, initial_size=8, device="cuda:0")
idx.train(xb)
idx.add(xb)
idx.n_probe = 4
v0, i0 = idx.search(xb[:, :64].contiguous(), k=5)
tpd.replicate_index(idx)                                 # ADVICE r5: raised "No backend type associated with device type cpu"
v1, i1 = idx.search(xb[:, :64].contiguous(), k=5)
g = tpd.init_groups(dev, want_rccl=True, timeout_s=120)  # on an NCCL-only default group: its own gloo control group
tpd.host_barrier(g.control)
out.update(same=bool(torch.equal(v0, v1) and torch.equal(i0, i1)), bulk_backend=g.bulk_backend,
           bulk_error=g.bulk_error, control=str(dist.get_backend(g.control)))
print(json.dumps(out))
dist.destroy_process_group()
'''


def test_replicate_index_on_an_nccl_only_default_group():
    """`init_process_group("nccl")` + `replicate_index(index)` -- the recipe INTEGRATION.md shows --
    has no CPU backend; the metadata broadcasts must not force device=cpu there, and init_groups must bring its own
    gloo control group"""
    port = 29800 + (os.getpid() % 2000)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", _NCCL_ONLY, ROOT, str(port)], env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["object_device"] == "None" and out["same"], out
    assert out["bulk_backend"] == "nccl" and out["bulk_error"] is None and "gloo" in out["control"], out


def test_rccl_itself_executes_with_one_rank():
    """The bulk plane of bench.py --gpus N is RCCL; a 1-GPU box cannot hold two RCCL ranks, but it can hold ONE: the
    group is created on the device, probed, and carries the chunked index broadcast, a 16-MiB broadcast and an
    all-reduce -- librccl loads, its kernels launch, HSA_ENABLE_IPC_MODE_LEGACY=0 is in effect."""
    port = 29700 + (os.getpid() % 2000)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, "-c", _RCCL_ONE, ROOT, str(port)], env=env, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert out["bulk_backend"] == "nccl", out
    assert out["bulk_error"] is None and out["same"] and out["replicated_bytes"] > 0
    assert out["all_reduce"] == [1.0, 1.0, 1.0, 1.0]
