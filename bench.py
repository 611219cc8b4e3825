#!/usr/bin/env python
"""bench.py -- queries/sec + recall@100 of IVFPQIndex.search on a SIFT1M-shaped index.

Headline workload (BASELINE.json configs[1]): d=128, 1 M base vectors, 100 k training vectors,
10 000 queries, IVFPQ n_cells=1024, m=64 (8-bit), n_probe=32, k=100, use_smart_probing=False
(deterministic scanned bytes).  With `--data-dir DIR` holding the TEXMEX files
(sift_base/learn/query.fvecs, sift_groundtruth.ivecs) the real SIFT1M is used; they are not
available offline, so the default is synthetic data of the same shape.

A "step" is one search() call over the whole resident query batch (coarse probe + LUT + list
scan + id map).  `value` = queries/s over all ranks; each rank owns a replica of the index (built
on rank 0, broadcast once over RCCL) and its own 10 000 queries -> weak scaling, no per-query
collective.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c4] [--no-secondary]

`--gpus N` with N > 1 launches its own N ranks (torch.distributed.run, one per GPU, RCCL) when
not already running under a launcher; under the driver's torchrun command it is one of the ranks.
`--scaling weak|strong` picks the mode `value` is quoted in (weak, the default: --nq queries per
rank; strong: ONE batch of --nq queries split over the ranks); at N > 1 the other mode is timed
as well and reported under `other_scaling`, with the per-rank ms_per_step spread and the bytes of
the one index broadcast.

At N=1 the headline is timed first (only the stream peak is measured ahead of it); then a time-boxed secondary pass
(<= ~60 s, configs[0] with its 80-s CPU leg last) puts the other BASELINE.json configs on the record
under the key "secondary" of the same JSON line, each with its own roofline:
  stream_peak  measured HBM streaming-read rate of this box (tpq_ubench_stream_read, 8 GiB)
  c3           GIST1M-shaped search() (d=960, m=120, n_probe=64, 1000 queries)
  c4           100 M-slot scan (n_cells=16384, m=64, n_probe=64; 6.4 GB of codes: the DRAM test)
  c5           MultiKMeans assign / update per Lloyd iteration (64 x 64 x 1 M, k=256)
  c1           configs[0]: the oracle's CPU train / add / search beside the same index on the GPU
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time


def rccl_env(env=os.environ):
    """What RCCL needs on this driver, set on EVERY entry path -- under the driver's own
    `torch.distributed.run ... bench.py --gpus N` as well as under self_launch -- and before the
    first HIP call (i.e. before torch is imported): the host driver only supports dmabuf IPC;
    without it RCCL's intra-node transport fails with `hipIpcGetMemHandle: invalid argument`."""
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    return env


rccl_env()

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0     # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 achievable
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 matrix-core peak (MI355X_MICROARCH.md)
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 matrix-core peak (MI355X_MICROARCH.md; not the 2:1-sparsity figure)
PROFILE_TAG = "r06"


# ---------------------------------------------------------------------------------------------
# data
# ---------------------------------------------------------------------------------------------
class SiftLike:
    """Synthetic stand-in for SIFT: non-negative, integer-valued fp32 in [0, 218].
    A vector = |blob centre + low-dimensional within-blob variation + isotropic noise|: the
    latent part gives the data a low intrinsic dimension (as real descriptors have), which is
    what makes an IVF probe of 32 / 1024 cells miss a few true neighbours -- recall@100 lands
    near the reference's 0.95 instead of a non-diagnostic 1.0."""

    def __init__(self, d, device, seed=1234, n_centers=256, latent_dim=10, latent_scale=55.0,
                 noise=12.0):
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        self.d, self.device = d, device
        self.noise, self.latent_scale = noise, latent_scale
        self.centers = torch.randn(d, n_centers, generator=g, device=device).abs() * 40.0
        self.basis = torch.randn(d, latent_dim, generator=g, device=device) / latent_dim ** 0.5

    def sample(self, n, seed):
        g = torch.Generator(device=self.device)
        g.manual_seed(seed)
        out = torch.empty(self.d, n, device=self.device, dtype=torch.float32)
        step = 1 << 18
        for b in range(0, n, step):
            e = min(n, b + step)
            a = torch.randint(0, self.centers.shape[1], (e - b,), generator=g, device=self.device)
            z = torch.randn(self.basis.shape[1], e - b, generator=g, device=self.device)
            x = self.centers[:, a] + (self.basis @ z) * self.latent_scale \
                + torch.randn(self.d, e - b, generator=g, device=self.device) * self.noise
            out[:, b:e] = x.abs().round().clamp_(0, 218)
        return out


def load_texmex(data_dir, name, device, n_base, nq):
    """(base, train, queries, groundtruth NN ids or None) as [d, n] device tensors"""
    from torchpq_amd import datasets
    paths = datasets.find_texmex(data_dir, name)
    if paths is None:
        return None
    base = torch.from_numpy(datasets.read_fvecs(paths["base"], n_base)).to(device).T.contiguous()
    learn = paths["learn"] and torch.from_numpy(datasets.read_fvecs(paths["learn"])).to(device).T.contiguous()
    query = torch.from_numpy(datasets.read_fvecs(paths["query"], nq)).to(device).T.contiguous()
    gt = None
    if paths["groundtruth"] and base.shape[1] == 1_000_000:
        gt = torch.from_numpy(datasets.read_ivecs(paths["groundtruth"], nq)[:, 0].astype(np.int64)).to(device)
    return base, learn, query, gt


def exact_nn(queries, base):
    """true nearest neighbour (L2) of every query column: GEMM + argmin, chunked"""
    b2 = (base * base).sum(0)
    out = []
    for q0 in range(0, queries.shape[1], 2048):
        d2 = b2[None, :] - 2.0 * (queries[:, q0:q0 + 2048].T @ base)
        out.append(d2.argmin(dim=1))
    return torch.cat(out)


# ---------------------------------------------------------------------------------------------
# index construction
# ---------------------------------------------------------------------------------------------
def build_index(args, device, base, train):
    from torchpq_amd.index import IVFPQIndex
    np.random.seed(1234)
    n_base = base.shape[1]
    idx = IVFPQIndex(d_vector=base.shape[0], n_subvectors=args.m, n_cells=args.n_cells,
                     initial_size=max(64, 2 * n_base // args.n_cells), device=str(device))
    torch.cuda.synchronize()
    t0 = time.time()
    idx.train(train)
    torch.cuda.synchronize()
    t_train = time.time() - t0
    t0 = time.time()
    for b in range(0, n_base, 1 << 18):
        idx.add(base[:, b:b + (1 << 18)].contiguous())
    idx.release_spare()  # the idle growth arena (as large as the index)
    torch.cuda.synchronize()
    return idx, t_train, time.time() - t0


def fabricate_index(device, d, m, n_cells, n_items, seed, slack=9):
    """configs[3] (SURVEY 8d "C4"): the codes are generated directly (no train/add): multinomial
    cell sizes, uniform random codes, random fp32 codebooks and coarse centroids; loaded through
    load_state_dict like any foreign index."""
    from torchpq_amd.index import IVFPQIndex
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    idx = IVFPQIndex(d_vector=d, n_subvectors=m, n_cells=n_cells, initial_size=1, device=str(device))
    probs = torch.full((n_cells,), 1.0 / n_cells, device=device)
    sizes = torch.zeros(n_cells, device=device, dtype=torch.long)
    for b in range(0, n_items, 1 << 24):  # multinomial(n_items, uniform) in draws of 16 M
        draw = torch.multinomial(probs, min(1 << 24, n_items - b), replacement=True, generator=g)
        sizes += torch.bincount(draw, minlength=n_cells)
    cap = sizes + slack
    start = torch.cumsum(cap, 0) - cap
    n_slots = int(cap.sum().item())
    storage = torch.empty(m // 4, n_slots, 4, device=device, dtype=torch.uint8)
    for gi in range(m // 4):  # in slices: randint draws int64 internally
        for s0 in range(0, n_slots, 1 << 26):
            e = min(n_slots, s0 + (1 << 26))
            storage[gi, s0:e] = torch.randint(0, 256, (e - s0, 4), generator=g, device=device,
                                              dtype=torch.uint8)
    pos = torch.arange(n_slots, device=device)
    cell_of = torch.repeat_interleave(torch.arange(n_cells, device=device), cap)
    occupied = (pos - start[cell_of]) < sizes[cell_of]
    a2i = torch.where(occupied, torch.cumsum(occupied.long(), 0) - 1, torch.full_like(pos, -1))
    del pos, cell_of
    sd = idx.state_dict()
    new = {
        "_storage": storage, "_cell_start": start, "_cell_size": sizes, "_cell_capacity": cap,
        "_is_empty": (~occupied).to(torch.uint8), "_address2id": a2i,
        "vq_codec._is_trained": torch.tensor(True, device=device),
        "pq_codec._is_trained": torch.tensor(True, device=device),
        "vq_codec.kmeans.centroids": torch.randn(d, n_cells, generator=g, device=device),
        "pq_codec.kmeans.centroids": torch.randn(m, d // m, 256, generator=g, device=device),
    }
    missing = set(sd) - set(new)
    for k in missing:
        new[k] = sd[k]
    idx.load_state_dict(new)
    return idx


# ---------------------------------------------------------------------------------------------
# measurement helpers
# ---------------------------------------------------------------------------------------------
def scanned_bytes(idx, queries, m):
    """sum over queries and probed cells of cell_size x m (uint8 codes only)"""
    tot = 0
    for q0 in range(0, queries.shape[1], 16384):
        _, cells, npl = idx.probe(queries[:, q0:q0 + 16384].contiguous())
        sizes = idx._cell_size[cells]
        live = torch.arange(cells.shape[1], device=cells.device)[None, :] < npl[:, None]
        tot += int((sizes * live).sum().item())
    return tot * m


def time_search(idx, queries, k, steps, warmup, barrier=None):
    """W untimed + K timed search() calls, barrier + synchronize on both sides; (wall seconds,
    mean scan-kernel ms per step from HIP events on the launch stream, query batches per step,
    last result, per-step statistics of the scan kernel)"""
    scan = idx._ivfpq_topk._scan
    for _ in range(warmup):
        idx.search(queries, k=k)
    torch.cuda.synchronize()
    if barrier is not None:
        barrier()
    scan.record_events = []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        vals, ids = idx.search(queries, k=k)
    torch.cuda.synchronize()
    if barrier is not None:
        barrier()
    dt = time.perf_counter() - t0
    events = scan.record_events
    scan.record_events = None
    nb = max(1, len(events) // max(steps, 1))  # query batches per search() call
    per_launch = [a.elapsed_time(b) for a, b in events]
    scan_ms = float(np.sum(per_launch)) / max(steps, 1) if events else float("nan")
    per_step = [float(np.sum(per_launch[i * nb:(i + 1) * nb])) for i in range(len(per_launch) // nb)]
    stats = {"n_split": scan.last_n_split, "queries_redone_exactly": None}
    if per_step:
        med = float(np.median(per_step))
        # one slow step (a clock dip, a page fault, another tenant) must be visible as such, not
        # averaged into the rate: the mean stays the quoted figure, the spread goes on the record
        stats.update({"kernel_ms_median": round(med, 4), "kernel_ms_min": round(min(per_step), 4),
                      "kernel_ms_max": round(max(per_step), 4),
                      "steps_slower_than_1p3x_median": int(sum(t > 1.3 * med for t in per_step))})
    # queries that took the exact redo (the candidate band overflowed, a table could not be scaled): one more,
    # untimed, search with the workspace kept; counted on every packed route (the one-launch finisher's own
    # redo branch, or the flag-gated exact kernel that ends the other routes) -- "slow and normally never taken"
    if nb == 1:  # (every packed route leaves its mark: IVFPQTopkHip.last_redone)
        scan.keep_workspace = True
        try:
            idx.search(queries, k=k)
            torch.cuda.synchronize()
            stats["queries_redone_exactly"] = scan.last_redone(queries.shape[1])
        finally:
            scan.keep_workspace = False
            scan.last_workspace = None
    return dt, scan_ms, nb, vals, ids, stats


INFINITY_CACHE_BYTES = 256 << 20  # MI355X_MICROARCH.md: 256 MiB MALL in front of the HBM stacks


def hbm_roofline(algo_bytes, kernel_ms, kernel, stream_peak=None, resident_bytes=None, stats=None, **extra):
    """`frac` per SURVEY 8(d): algorithmic bytes / kernel time / 8 TB/s.  `fed_by` names the level that
    can have fed it: an array that fits the 256 MiB Infinity Cache is served from there across steps
    (its rate may exceed the DRAM stream peak and says nothing about HBM); the DRAM evidence is the
    100 M-slot workload (c4)."""
    achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "kernel": kernel,
         "kernel_ms": round(kernel_ms, 4), "algorithmic_bytes_per_launch": int(algo_bytes)}
    if stream_peak:
        r["stream_peak"] = round(stream_peak, 1)
        r["frac_of_stream_peak"] = round(achieved / stream_peak, 4)
    if resident_bytes is not None:
        # `fed_by` is decided by what was MEASURED where a measurement can decide it: a rate above the
        # box's DRAM stream peak cannot have come from DRAM alone; below it, an array that fits the 256 MiB MALL and is
        # re-read many times per launch is served from there after its first touch; an array beyond the MALL read at or
        # below the stream peak is DRAM-fed (the 100 M-slot record adds the "cold" variant: every byte once per launch)
        r["code_bytes_resident"] = int(resident_bytes)
        r["reads_of_each_code_byte_per_launch"] = round(algo_bytes / max(resident_bytes, 1), 1)
        if stream_peak and achieved > stream_peak:
            r["fed_by"] = "infinity_cache" if resident_bytes <= 2 * INFINITY_CACHE_BYTES else "hbm+infinity_cache"
            r["fed_by_basis"] = (f"measured: {achieved:.0f} GB/s is above this box's DRAM stream peak "
                                 f"({stream_peak:.0f} GB/s), so part of it is re-reads served by the Infinity Cache")
        elif resident_bytes <= INFINITY_CACHE_BYTES:
            r["fed_by"] = "infinity_cache"
            r["fed_by_basis"] = "the code array fits the 256 MiB Infinity Cache and every byte is re-read within a launch"
        else:
            r["fed_by"] = "hbm"
            r["fed_by_basis"] = "code array beyond the Infinity Cache, rate at or below the DRAM stream peak"
    if stats:
        r.update(stats)
    r.update(extra)
    return r


def cross_check_profile(roofline, name, kernel_prefix):
    """the tracked rocprofv3 --kernel-trace --stats summary of the same workload: its average duration
    of the dominant kernel must agree with this run's HIP-event figure; > 10 % apart is flagged on the
    line itself (`profile_mismatch`), so a record and the profile quoted for it cannot silently differ"""
    prof = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{name}.json")
    try:
        pj = json.load(open(prof))
        rows = [k for k in pj.get("kernels", []) if kernel_prefix in k["name"]]
        if not rows:
            return
        top = max(rows, key=lambda k: k["avg_us"] * k["calls"])
        launches = max(1, int(roofline.get("launches_per_step", 1)))
        # the HIP events of this run bracket the whole scan call: the dominant kernel plus, on the large-batch route
        # at m = 64, its finish kernel (one launch each per call)
        tail = [k for k in pj.get("kernels", []) if "scan_finish_exact_kernel" in k["name"]
                and k["calls"] == top["calls"]] if ", -1" in top["name"] else []
        per_call_us = top["avg_us"] + sum(k["avg_us"] for k in tail)
        roofline["kernel_ms_profile"] = round(per_call_us * launches / 1e3, 4)
        roofline["kernel_ms_profile_dominant"] = round(top["avg_us"] * launches / 1e3, 4)
        roofline["kernel_ms_profile_source"] = (f"profiles/{PROFILE_TAG}_{name}.json ({top['calls']} calls of "
                                                f"{top['name'][:60]}" + (" + its finish kernel" if tail else "") + ")")
        roofline["profile_mismatch"] = bool(
            abs(roofline["kernel_ms"] / roofline["kernel_ms_profile"] - 1.0) > 0.10)
    except Exception as e:  # a missing / malformed summary must not break the bench line
        roofline["kernel_ms_profile"] = None
        roofline["profile_note"] = f"{type(e).__name__}: {e}"[:200]


def source_fingerprint():
    """sha1 over the kernel sources with comments and whitespace removed: a committed PMC summary is
    attached to the bench line only when it was taken from exactly this code (editing a comment
    does not orphan the profiles)"""
    import re
    h = hashlib.sha1()
    src = os.path.join(ROOT, "torchpq_amd", "csrc")
    for f in sorted(os.listdir(src)):
        if f.endswith((".h", ".hip", ".cpp")):
            text = open(os.path.join(src, f), encoding="utf-8", errors="replace").read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            text = re.sub(r"//[^\n]*", "", text)
            h.update(f.encode())
            h.update(re.sub(r"\s+", "", text).encode())
    return h.hexdigest()[:16]


def attach_traffic(roofline, name):
    """HBM-side bytes per launch come from a separate rocprofv3 --pmc pass over the same command
    (FETCH_SIZE, corrected as MI355X_MICROARCH.md prescribes; tools/profile_bench.sh); the
    committed summary is read back only if its source fingerprint and byte count match"""
    prof = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{name}.json")
    if not os.path.exists(prof):
        return
    try:
        pj = json.load(open(prof))
        algo = roofline.get("algorithmic_bytes_per_launch")
        if pj.get("source_fingerprint") != source_fingerprint():
            roofline["traffic_note"] = f"profiles/{PROFILE_TAG}_{name}.json is from other kernel sources: not attached"
            return
        if algo is None:  # compute-bound kernel (flops roofline): the HBM bytes it read, as measured
            roofline["traffic"] = round(pj["hbm_side_read_bytes_corrected"])
            roofline["traffic_source"] = (f"profiles/{PROFILE_TAG}_{name}.json (rocprofv3 --pmc "
                                          "FETCH_SIZE x2 x1KiB, same command, same sources)")
            roofline["traffic_measured_in_this_run"] = False
        elif abs(pj["algorithmic_bytes_per_launch"] - algo) <= 0.02 * algo:
            roofline["traffic"] = round(pj["hbm_side_read_bytes_corrected"])
            roofline["traffic_source"] = (f"profiles/{PROFILE_TAG}_{name}.json (rocprofv3 --pmc "
                                          "FETCH_SIZE x2 x1KiB, same command, same sources)")
            # replayed from the builder's committed counter pass (another run, possibly another box
            # of the pool): PMC counters cannot be collected from inside the timed run
            roofline["traffic_measured_in_this_run"] = False
    except Exception as e:  # a malformed summary must not break the bench line
        roofline["traffic_note"] = f"unreadable profile: {e}"


TRAFFIC_PASS = {"enabled": True}


def pmc_traffic(child_args, kernel_filter, timeout_s=240, grid_threads=None):
    """HBM-side read bytes per launch of `kernel_filter`, measured IN THIS RUN: the same workload once more in a
    child process under `rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum` (a counter pass of its own: PMC
    collection serialises the kernels and cannot share a process with the timed region), corrected as
    MI355X_MICROARCH.md prescribes (FETCH_SIZE is in KiB and counts 64 B per 128-B request on gfx950: x 1024
    x 2).  None when the pass is switched off, rocprofv3 is missing, this process itself runs under a
    profiler, or the child fails / overruns -- the caller then falls back to the committed profile."""
    import csv
    import glob
    import shutil
    import signal
    import tempfile
    if not TRAFFIC_PASS["enabled"] or os.environ.get("TPQ_BENCH_NO_PMC", "0") == "1":
        return None
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
        return None
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    try:
        td = tempfile.mkdtemp(prefix="tpq_pmc_", dir="/tmp")
    except OSError:
        return None
    env = dict(os.environ, TMPDIR="/tmp")
    cmd = [exe, "--pmc", "FETCH_SIZE", "TCC_EA0_RDREQ_sum", "--output-format", "csv", "-d", td, "-o", "run", "--",
           sys.executable, os.path.abspath(__file__)] + list(child_args) + ["--no-traffic-pass"]
    t0 = time.time()
    try:
        proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                start_new_session=True)
        try:
            proc.wait(timeout=timeout_s)
        except subprocess.TimeoutExpired:
            os.killpg(proc.pid, signal.SIGKILL)   # (its own session: the exact process group started above)
            proc.wait()
            return None
        fetch, rdreq = [], []
        for f in glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kernel_filter in r["Kernel_Name"]:
                    # (`grid_threads`: only the launches of that many threads -- the child of the 100 M-slot record
                    # also runs the cold variant through the same kernel, 2 048 workgroups instead of 10 000)
                    if grid_threads is not None and int(r.get("Grid_Size", -1)) != grid_threads:
                        continue
                    if r["Counter_Name"] == "FETCH_SIZE":
                        fetch.append(float(r["Counter_Value"]))
                    elif r["Counter_Name"] == "TCC_EA0_RDREQ_sum":
                        rdreq.append(float(r["Counter_Value"]))
        if not fetch:
            return None
        out = {"traffic": round(float(np.mean(fetch)) * 1024 * 2), "launches": len(fetch),
               "pass_s": round(time.time() - t0, 1)}
        if rdreq:
            out["tcc_ea_rdreq_x128B"] = round(float(np.mean(rdreq)) * 128)
        return out
    except Exception:  # noqa: BLE001 -- the counter pass must never break the bench line
        return None
    finally:
        shutil.rmtree(td, ignore_errors=True)


def measured_traffic(roofline, child_args, kernel_filter, grid_threads=None):
    """roofline["traffic"] from this run's own counter pass; True when it was attached"""
    t = pmc_traffic(child_args, kernel_filter, grid_threads=grid_threads)
    if not t:
        return False
    algo = roofline.get("algorithmic_bytes_per_launch")
    roofline["traffic"] = t["traffic"]
    roofline["traffic_measured_in_this_run"] = True
    roofline["traffic_source"] = (f"rocprofv3 --pmc FETCH_SIZE (KiB, x2 on gfx950) in a child pass of this run over "
                                  f"the same workload: mean of {t['launches']} launches of {kernel_filter}, "
                                  f"{t['pass_s']} s")
    if "tcc_ea_rdreq_x128B" in t:
        roofline["traffic_tcc_ea_rdreq_x128B"] = t["tcc_ea_rdreq_x128B"]
    if algo:
        roofline["traffic_over_algorithmic"] = round(t["traffic"] / algo, 4)
    return True


STREAM_SETTINGS = (
    # (label, n_blocks per CU, threads, unroll, chunk_bytes, nontemporal); chunk 0 = the buffer cut evenly
    ("grid-stride, 8 x 256 threads per CU, 8 in flight, nt (round 1-5's figure)", None, 256, 8, 0, 1),
    ("even cut, 8 x 256 per CU, 8 in flight, nt", 8, 256, 8, 0, 1),
    ("even cut, 8 x 256 per CU, 16 in flight, nt", 8, 256, 16, 0, 1),
    ("even cut, 4 x 512 per CU, 8 in flight, plain loads", 4, 512, 8, 0, 0),
    ("cells of 390 592 B (a C4 cell) dealt to 4 x 256 per CU, 4 in flight, plain loads (the scan's pattern)",
     4, 256, 4, 390592, 0),
    ("cells of 390 592 B dealt to 4 x 256 per CU, 8 in flight, nt", 4, 256, 8, 390592, 1),
    ("cells of 62 528 B (a C2 cell) dealt to 8 x 256 per CU, 8 in flight, nt", 8, 256, 8, 62528, 1),
    ("2 MiB pieces dealt to 2 x 1024 per CU, 8 in flight, nt", 2, 1024, 8, 2 << 20, 1),
)


def stream_peak_sweep(device, gib=8, iters=3):
    """sustained HBM read rate of the box: the streaming-read microkernel over a buffer far beyond the 256 MiB
    Infinity Cache (every byte read once per launch: DRAM, not cache), in a handful of settings -- the scan's own
    access pattern among them --, best of `iters` each (HIP events on the launch stream).  Returns (best GB/s,
    [(label, GB/s)])."""
    from torchpq_amd import _lib
    lib = _lib.load()
    buf = torch.empty(gib << 30, device=device, dtype=torch.uint8)
    buf.view(torch.int64).fill_(0x0123456789abcdef)
    sink = torch.zeros(1, device=device, dtype=torch.int32)
    st = _lib.stream_ptr(device)
    cus = torch.cuda.get_device_properties(device).multi_processor_count
    rows = []
    with torch.cuda.device(device):
        for label, per_cu, threads, unroll, chunk, nt in STREAM_SETTINGS:
            best = 0.0
            for it in range(iters + 1):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if per_cu is None:
                    _lib.check(lib.tpq_ubench_stream_read(_lib.ptr(buf), buf.numel(), _lib.ptr(sink), 0, st),
                               "tpq_ubench_stream_read")
                else:
                    _lib.check(lib.tpq_ubench_stream_read_ex(_lib.ptr(buf), buf.numel(), _lib.ptr(sink), per_cu * cus,
                                                             threads, unroll, chunk, nt, st),
                               "tpq_ubench_stream_read_ex")
                e1.record()
                torch.cuda.synchronize()
                if it:  # first launch = warm-up
                    best = max(best, buf.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9)
            rows.append((label, round(best, 1)))
    return max(r[1] for r in rows), rows


def stream_peak_gbps(device, gib=8, iters=3):
    return stream_peak_sweep(device, gib, iters)[0]


def stream_peak_record(device):
    sp, rows = stream_peak_sweep(device)
    return sp, {"value": round(sp, 1), "unit": "GB/s", "frac_of_spec": round(sp / HBM_PEAK_GBPS, 4),
                "what": "tpq_ubench_stream_read[_ex], 8 GiB buffer read once per launch (DRAM: 32 x the Infinity Cache), "
                        "16-byte loads, best setting, best of 3",
                "settings_GBps": {label: v for label, v in rows}}


def cpu_baseline(idx, queries, k, n_sample):
    """The oracle (the CPU restatement of the reference's algorithm) on a bounded sample, all host cores, stage by
    stage: coarse step (BLAS GEMM + the 2ab - a^2 - b^2 epilogue + row top-n_probe, numpy), ADC LUT
    (C / OpenMP fma chains), list scan + top-k (C / OpenMP), address -> id."""
    from oracle import c_oracle
    from oracle import ivfpq_oracle as orc
    x = queries[:, :n_sample].cpu().numpy()
    vq = idx.vq_codec.codebook.cpu().numpy()
    pq = idx.pq_codec.codebook.cpu().numpy()
    storage = idx._storage.cpu().numpy()
    is_empty = idx._is_empty.cpu().numpy()
    cs, sz = idx._cell_start.cpu().numpy(), idx._cell_size.cpu().numpy()
    a2i = idx._address2id.cpu().numpy()
    cores = os.cpu_count() or 1
    try:
        torch.set_num_threads(cores)   # (numpy's BLAS follows its own environment; logged below)
    except RuntimeError:
        pass
    t0 = time.time()
    sims = orc.neg_sq_l2(x, vq)
    topk_sims, cells = orc.topk_desc(sims, idx.n_probe)
    if idx.use_smart_probing and idx.n_probe > 1:
        npl = orc.smart_probing(topk_sims, idx.n_probe, idx.smart_probing_temperature)
    else:
        npl = np.full(x.shape[1], idx.n_probe, dtype=np.int64)
    t1 = time.time()
    lut = c_oracle.adc_lut(x, pq, idx.distance, n_threads=cores)
    t2 = time.time()
    vals, adr = c_oracle.scan_topk(storage, lut, is_empty, cs[cells], sz[cells], npl, k, n_threads=cores)
    ids = orc.get_id_by_address(a2i, adr)
    t3 = time.time()
    dt = t3 - t0
    return {"value": round(n_sample / dt, 1), "unit": "queries/s", "cores": cores, "kind": "port",
            "split_s": {"coarse": round(t1 - t0, 3), "lut": round(t2 - t1, 3), "scan": round(t3 - t2, 3)},
            "sample": f"{n_sample} of the {queries.shape[1]} queries, full pipeline (coarse: numpy BLAS GEMM + "
                      f"epilogue + top-n_probe; LUT and list scan: C/OpenMP on {cores} threads), {dt:.1f} s"}, ids


def oracle_sample_check(idx, queries, vals, ids, k, n_sample=32, host=None, cells=None, npl=None):
    """The TIMED route checked where it was timed: `n_sample` queries spread over the batch
    that was just searched (first and last included), the C oracle (adc_lut + scan_topk: the reference's
    arithmetic, ivfpq_topk.cu:822-971) run on the GPU's own probed cells of exactly those queries, and the
    rows the timed call returned compared with it -- values bit for bit, ids entry by entry.  `host`: a dict
    that caches the index's host copies between calls on the same index."""
    from oracle import c_oracle
    from oracle import ivfpq_oracle as orc
    t0 = time.time()
    nq = queries.shape[1]
    n_sample = max(1, min(n_sample, nq))
    sel = torch.unique(torch.linspace(0, nq - 1, n_sample, device=queries.device).round().long())
    qs = queries[:, sel].contiguous()
    if idx.distance == "cosine":
        from torchpq_amd import util
        qs = util.normalize(qs, dim=0)
    if cells is None:
        _, cells, npl = idx.probe(qs)
    else:  # (search_cells with the caller's cells: no coarse step to repeat)
        cells, npl = cells[sel], npl[sel]
    cells_h, npl_h = cells.cpu().numpy(), npl.cpu().numpy()
    host = host if host is not None else {}
    if host.get("version") != (idx._storage.data_ptr(), idx.n_items):
        host.clear()
        host.update(version=(idx._storage.data_ptr(), idx.n_items), storage=idx._storage.cpu().numpy(),
                    is_empty=idx._is_empty.cpu().numpy(), a2i=idx._address2id.cpu().numpy(),
                    cs=idx._cell_start.cpu().numpy(), sz=idx._cell_size.cpu().numpy(),
                    pq=idx.pq_codec.codebook.cpu().numpy())
    lut = c_oracle.adc_lut(qs.cpu().numpy(), host["pq"], idx.distance)
    ev, ea = c_oracle.scan_topk(host["storage"], lut, host["is_empty"], host["cs"][cells_h], host["sz"][cells_h],
                                npl_h, k)
    ei = orc.get_id_by_address(host["a2i"], ea)
    gv, gi = vals[sel].cpu().numpy(), ids[sel].cpu().numpy()
    return {"queries_checked": int(sel.numel()), "of_the_timed_batch_of": int(nq),
            "ids_equal_to_oracle": round(float((gi == ei).mean()), 6),
            "values_bit_equal": bool(np.array_equal(gv.view(np.int32), ev.view(np.int32))),
            "rows_fully_equal": int(((gi == ei).all(1) & (gv.view(np.int32) == ev.view(np.int32)).all(1)).sum()),
            "recall_vs_ref": round(float(np.mean([len(np.intersect1d(gi[r], ei[r])) for r in range(gi.shape[0])]))
                                   / k, 4),
            "max_address_checked": int(ea.max()), "addresses_beyond_2p24": int((ea >= (1 << 24)).sum()),
            "check_s": round(time.time() - t0, 2),
            "what": "rows of the TIMED search() call vs oracle/ (C adc_lut + scan_topk on the GPU's probed cells of "
                    "the same queries, address -> id)"}


def ids_digest(vals, ids):
    """sha1 over the raw bytes of a result (values, then ids): equal digests <=> bit-equal results"""
    h = hashlib.sha1()
    h.update(vals.contiguous().cpu().numpy().tobytes())
    h.update(ids.contiguous().cpu().numpy().tobytes())
    return h.hexdigest()[:16]


# ---------------------------------------------------------------------------------------------
# secondary pass: BASELINE.json configs[2..4] + the box's stream peak (N=1, rank 0, time-boxed)
# ---------------------------------------------------------------------------------------------
def secondary_c3(device, stream_peak, steps=20):
    """configs[2]: GIST1M-shaped index built through train/add on synthetic d=960 data in [0,1]"""
    from torchpq_amd.index import IVFPQIndex
    d, m, n_cells, n, nq, n_probe, k = 960, 120, 1024, 1_000_000, 1000, 64, 100
    g = torch.Generator(device=device)
    g.manual_seed(1235)
    centers = torch.rand(d, 512, generator=g, device=device)
    basis = torch.randn(d, 24, generator=g, device=device) / 24 ** 0.5

    def sample(count):
        a = torch.randint(0, 512, (count,), generator=g, device=device)
        z = torch.randn(24, count, generator=g, device=device)
        x = centers[:, a] * 0.6 + (basis @ z) * 0.12 + torch.randn(d, count, generator=g, device=device) * 0.04
        return x.clamp_(0, 1)

    np.random.seed(1235)
    idx = IVFPQIndex(d_vector=d, n_subvectors=m, n_cells=n_cells, initial_size=2 * n // n_cells,
                     device=str(device))
    t0 = time.time()
    idx.train(sample(100_000))
    torch.cuda.synchronize()
    t_train = time.time() - t0
    t0 = time.time()
    for b in range(0, n, 1 << 17):
        idx.add(sample(min(1 << 17, n - b)))
    idx.release_spare()
    torch.cuda.synchronize()
    t_add = time.time() - t0
    idx.n_probe, idx.use_smart_probing = n_probe, False
    queries = sample(nq)
    dt, scan_ms, nb, vals, ids, stats = time_search(idx, queries, k, steps, 3)
    algo = scanned_bytes(idx, queries, m)
    return {"workload": f"GIST1M-like d={d} n={n} IVFPQ n_cells={n_cells} m={m} nprobe={n_probe} k={k}, "
                        f"{nq} queries, search() end to end",
            "value": round(nq * steps / dt, 1), "unit": "queries/s", "ms_per_step": round(dt / steps * 1e3, 4),
            "train_s": round(t_train, 2), "add_s": round(t_add, 2),
            "oracle_check": oracle_sample_check(idx, queries, vals, ids, k),
            "roofline": hbm_roofline(algo, scan_ms, "scan_packed_kernel<1,120,false,2> (one launch: scan + merge + "
                                     "write)", stream_peak, resident_bytes=idx._storage.numel(), stats=stats,
                                     launches_per_step=nb, bytes_per_query=round(algo / nq, 1))}


def secondary_c4(device, stream_peak, steps=5):
    """configs[3] on one GPU: 100 M slots, 6.4 GB of codes -- beyond the Infinity Cache"""
    d, m, n_cells, n, nq, n_probe, k = 128, 64, 16384, 100_000_000, 10000, 64, 100
    idx = fabricate_index(device, d, m, n_cells, n, seed=1236)
    idx.n_probe, idx.use_smart_probing = n_probe, False
    g = torch.Generator(device=device)
    g.manual_seed(4236)
    queries = torch.randn(d, nq, generator=g, device=device)
    if C4_VARIANT["mode"] == "cold":  # (profiling / counter passes of the cold variant alone)
        return {"workload": "the cold variant of configs[3] alone", "cold": c4_cold_variant(idx, device, stream_peak, k, steps, {}),
                "roofline": {}}
    dt, scan_ms, nb, vals, ids, stats = time_search(idx, queries, k, steps, 1)
    algo = scanned_bytes(idx, queries, m)
    host = {}
    rec = {"workload": f"synthetic codes d={d} n={n} IVFPQ n_cells={n_cells} m={m} nprobe={n_probe} k={k}, "
                       f"{nq} queries per GPU, search() end to end (coarse probe + fused LUT + scan)",
           "value": round(nq * steps / dt, 1), "unit": "queries/s", "ms_per_step": round(dt / steps * 1e3, 4),
           "oracle_check": oracle_sample_check(idx, queries, vals, ids, k, host=host),
           "roofline": hbm_roofline(algo, scan_ms, "scan_packed_kernel<1,64,false,-16> (four-wave workgroups over the "
                                    "16-bit selection table) + scan_finish_exact_kernel; kernel_ms brackets both",
                                    stream_peak, resident_bytes=idx._storage.numel(), stats=stats,
                                    launches_per_step=nb, bytes_per_query=round(algo / nq, 1))}
    if C4_VARIANT["mode"] == "warm":  # (profiling / counter passes: per-kernel averages of the timed batch alone)
        return rec
    try:
        rec["cold"] = c4_cold_variant(idx, device, stream_peak, k, steps, host)
        rec["roofline"]["dram_frac"] = rec["cold"]["roofline"]["frac"]
        rec["roofline"]["dram_frac_what"] = ("the same kernels on the COLD variant of this index (`cold`): every cell "
                                             "probed by exactly one query, each code byte read once per launch")
        warm, cold = rec["roofline"]["achieved"], rec["cold"]["roofline"]["achieved"]
        rec["roofline"]["rate_over_cold_rate"] = round(warm / cold, 4)
    except Exception as e:  # noqa: BLE001 -- the variant must not cost the record
        rec["cold"] = {"error": repr(e)[:300]}
    return rec


C4_VARIANT = {"mode": "both"}   # --c4-variant: "warm" / "cold" run one half of the record (profiles, counter passes)
C4_COLD_QUERIES = 2048  # (two rounds of the chip's 1 024 workgroup slots: 1 024 x 16 / 2 048 x 8 / 4 096 x 4 / 8 192 x 2
#                          measured 5.19 / 5.49 / 5.40 / 5.05 TB/s on one box, tools/ab_stream.py)


def c4_cold_variant(idx, device, stream_peak, k, steps, host):
    """how much of the 100 M-slot scan's rate is DRAM?  In the timed batch each cell is probed by ~39 of
    the 10 000 queries, and re-reads that land within the Infinity Cache's reach are counted by FETCH_SIZE like DRAM
    reads.  Here the SAME index is searched with the 16 384 cells dealt to 2 048 queries, 8 each, every cell exactly once
    (a random permutation): each of the 6.4 GB of code bytes is read ONCE per launch, 25 x the Infinity Cache apart -- what
    this runs at is DRAM.  2 048 queries take the large-batch route the headline takes (four-wave workgroups, 16-bit
    selection table, finish kernel)."""
    n_cells, m = idx.n_cells, idx.n_subvectors
    nq, n_probe = C4_COLD_QUERIES, n_cells // C4_COLD_QUERIES
    g = torch.Generator(device=device)
    g.manual_seed(977)
    cells = torch.randperm(n_cells, generator=g, device=device).view(nq, n_probe).contiguous()
    queries = torch.randn(idx.d_vector, nq, generator=g, device=device)
    npl = torch.full((nq,), n_probe, device=device, dtype=torch.long)
    scan = idx._ivfpq_topk._scan
    run = lambda: idx.search_cells(queries, cells, n_probe_list=npl, k=k)  # noqa: E731
    steps = max(steps, 20)   # (1.2 ms a step: the spread of five was 1.22 ... 1.40)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    scan.record_events = []
    t0 = time.perf_counter()
    for _ in range(steps):
        vals, ids = run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    per = [a.elapsed_time(b) for a, b in scan.record_events]
    scan.record_events = None
    algo = int(idx._cell_size.sum().item()) * m
    stats = {"kernel_ms_min": round(min(per), 4), "kernel_ms_median": round(float(np.median(per)), 4),
             "kernel_ms_max": round(max(per), 4), "n_split": scan.last_n_split}
    roof = hbm_roofline(algo, float(np.mean(per)), "scan_packed_kernel<1,64,false,-16> + scan_finish_exact_kernel",
                        stream_peak, resident_bytes=idx._storage.numel(), stats=stats)
    roof["fed_by"], roof["fed_by_basis"] = "hbm", "by construction: every code byte is read once per launch"
    return {"workload": f"the same index, {nq} queries x {n_probe} cells, every cell probed exactly once per launch "
                        f"(search_cells: fused LUT + scan + finish; no coarse step)",
            "value": round(nq * steps / dt, 1), "unit": "queries/s", "ms_per_step": round(dt / steps * 1e3, 4),
            "oracle_check": oracle_sample_check(idx, queries, vals, ids, k, host=host, cells=cells, npl=npl),
            "roofline": roof}


def secondary_residual(device, stream_peak, steps=10):
    """residual PQ (`pq_use_residual=True`, ivfpq_topk.cu:1039-1208 /
    index/IVFPQIndex.py:366-450) at the configs[1] shape, search() end to end; algorithmic bytes per slot = m code
    bytes + the 4-byte per-slot term the packed residual scan reads beside them."""
    from oracle import c_oracle
    from oracle import ivfpq_oracle as orc
    from torchpq_amd.index import IVFPQIndex
    d, m, n_cells, n, nq, n_probe, k = 128, 64, 1024, 1_000_000, 10000, 32, 100
    synth = SiftLike(d, device)
    base = synth.sample(n, seed=1)
    g = torch.Generator(device=device)
    g.manual_seed(2)
    train = base[:, torch.randperm(n, generator=g, device=device)[:100_000]].contiguous()
    np.random.seed(1234)
    idx = IVFPQIndex(d_vector=d, n_subvectors=m, n_cells=n_cells, initial_size=2 * n // n_cells, device=str(device),
                     pq_use_residual=True)
    t0 = time.time()
    idx.train(train)
    torch.cuda.synchronize()
    t_train = time.time() - t0
    t0 = time.time()
    for b in range(0, n, 1 << 18):
        idx.add(base[:, b:b + (1 << 18)].contiguous())
    idx.release_spare()
    torch.cuda.synchronize()
    t_add = time.time() - t0
    del base, train
    idx.n_probe, idx.use_smart_probing = n_probe, False
    queries = synth.sample(nq, seed=4321)
    dt, scan_ms, nb, vals, ids, stats = time_search(idx, queries, k, steps, 2)
    slots = scanned_bytes(idx, queries, m) // m
    algo = slots * (m + 4)
    # oracle sample: the GPU's probed cells and base sims, the index's own part1 / part2 tables (the kernel's
    # arithmetic; part2 is a library GEMM), the C restatement of ivfpq_topk_residual_precomputed over them
    sel = torch.unique(torch.linspace(0, nq - 1, 32, device=device).round().long())
    qs = queries[:, sel].contiguous()
    topk_sims, cells, npl = idx.probe(qs)
    p1, p2 = idx.precomputed_adc_residual_precomputed(qs)
    N = lambda t: t.cpu().numpy()  # noqa: E731
    cells_h = N(cells)
    ev, ea = c_oracle.scan_topk_residual(N(idx._storage), N(p1), N(p2.contiguous()), cells_h, N(topk_sims),
                                         N(idx._is_empty), N(idx._cell_start)[cells_h], N(idx._cell_size)[cells_h],
                                         N(npl), k)
    ei = orc.get_id_by_address(N(idx._address2id), ea)
    gv, gi = N(vals[sel]), N(ids[sel])
    return {"workload": f"residual PQ (pq_use_residual=True), SIFT1M-like d={d} n={n} n_cells={n_cells} m={m} "
                        f"nprobe={n_probe} k={k}, {nq} queries, search() end to end",
            "value": round(nq * steps / dt, 1), "unit": "queries/s", "ms_per_step": round(dt / steps * 1e3, 4),
            "train_s": round(t_train, 2), "add_s": round(t_add, 2),
            "oracle_check": {"queries_checked": int(sel.numel()), "ids_equal_to_oracle": round(float((gi == ei).mean()), 6),
                             "values_bit_equal": bool(np.array_equal(gv.view(np.int32), ev.view(np.int32))),
                             "what": "rows of the TIMED search() call vs oracle/ scan_topk_residual (C) on the GPU's "
                                     "probed cells, base sims and part1 / part2 tables"},
            "roofline": hbm_roofline(algo, scan_ms, "scan_packed_kernel<.,64,RES=true> + scan_merge_refine_kernel + "
                                     "the flag-gated scan_residual_kernel; kernel_ms brackets the scan call",
                                     stream_peak, resident_bytes=idx._storage.numel() + 4 * idx._storage.shape[1],
                                     stats=stats, launches_per_step=nb, bytes_per_slot=m + 4,
                                     bytes_per_query=round(algo / nq, 1))}


def secondary_flat(device, stream_peak, steps=5):
    """FlatIndex (exact search, index/FlatIndex.py:44-101) -- one library GEMM + the HIP row top-k --, the
    recall ground-truth generator, at the configs[1] base size.  Bound: HBM -- the [nq, n] similarity matrix is written
    once and read once (8 B per (query, vector) pair); the GEMM's 2 nq n d flop ride beneath it."""
    from torchpq_amd.index import FlatIndex
    d, n, nq, k = 128, 1_000_000, 1000, 100
    synth = SiftLike(d, device)
    base = synth.sample(n, seed=1)
    idx = FlatIndex(d_vector=d, initial_size=n, device=str(device))
    idx.add(base)
    queries = synth.sample(nq, seed=4321)
    idx.search(queries, k=k)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        vals, ids = idx.search(queries, k=k)
    e1.record()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = e0.elapsed_time(e1) / steps
    # oracle sample: exact brute force in float64 on the host for 16 queries
    sel = np.unique(np.linspace(0, nq - 1, 16).round().astype(np.int64))
    xb = base.cpu().numpy().astype(np.float64)
    xq = queries[:, torch.from_numpy(sel).to(device)].cpu().numpy().astype(np.float64)
    sims = 2.0 * (xq.T @ xb) - (xq * xq).sum(0)[:, None] - (xb * xb).sum(0)[None, :]
    order = np.argsort(-sims, axis=1, kind="stable")[:, :k]
    ev = np.take_along_axis(sims, order, 1)
    gv, gi = vals.cpu().numpy()[sel], ids.cpu().numpy()[sel]
    overlap = float(np.mean([len(np.intersect1d(gi[r], order[r])) for r in range(len(sel))])) / k
    rel = float(np.max(np.abs(gv - ev) / np.maximum(np.abs(ev), 1.0)))
    algo = 8 * nq * n
    roof = hbm_roofline(algo, ms, "rocBLAS SGEMM (library) + topk_select_kernel (HIP) + id_by_address; kernel_ms = the "
                                  "whole search() on the stream", stream_peak)
    roof["gemm_flops_per_launch"] = 2.0 * nq * n * d
    return {"workload": f"FlatIndex exact search d={d} n={n}, {nq} queries, k={k} (SIFT1M-like)",
            "value": round(nq * steps / dt, 1), "unit": "queries/s", "ms_per_step": round(dt / steps * 1e3, 4),
            "oracle_check": {"queries_checked": int(len(sel)), "top_k_overlap_with_float64_brute_force": round(overlap, 4),
                             "values_max_rel_diff": rel,
                             "what": "float64 brute force on the host; fp32 GEMM summation order differs from the exact "
                                     "order, so ids are compared as sets and values at 1e-4 (BASELINE.json tolerance)"},
            "roofline": roof}


def secondary_c1(device, stream_peak):
    """configs[0] ("plumbing, no GPU"): d=128 n=100k n_cells=256 m=16 n_probe=8 k=10, x ~ N(0,1)
    (SURVEY 8d).  CPU leg = the ORACLE's train / add / search -- the restated reference CPU path
    (KMeans.fit + get_labels + compute_centroids, clustering/KMeans.py:323-438; CellContainer.add,
    container/CellContainer.py:313-367; the search restatement) -- timed on the host cores.  GPU leg =
    that very index (the oracle's codebooks, codes and cell table, loaded through load_state_dict) searched
    by IVFPQIndex.search(), ids / values compared; and the same data trained / added / searched by the
    GPU path for the rates."""
    from oracle import c_oracle
    from oracle import ivfpq_oracle as orc
    from torchpq_amd.index import IVFPQIndex
    d, n, nq, n_cells, m, n_probe, k = 128, 100_000, 1000, 256, 16, 8, 10
    cores = os.cpu_count() or 1
    g = torch.Generator()
    g.manual_seed(0)
    x = torch.randn(d, n, generator=g).numpy()
    q = torch.randn(d, nq, generator=g).numpy()

    def assign(a, b):
        return c_oracle.max_sim(a, b, "euclidean", "direct", n_threads=cores)

    # ---- CPU leg (the oracle) ----
    np.random.seed(0)
    t0 = time.time()
    vq, _, _, vq_steps = orc.kmeans_fit_redo(x[None], None, 1, 15, 1e-4, n_cells, assign=assign)
    xs = np.ascontiguousarray(x.reshape(m, d // m, n))
    pq, _, _, pq_steps = orc.kmeans_fit_redo(xs, None, 1, 25, 1e-4, 256, assign=assign)
    t1 = time.time()
    cells = assign(x[None], vq)[1][0]
    codes = assign(xs, pq)[1].astype(np.uint8)
    cont = orc.ContainerState(code_size=m, n_cells=n_cells, initial_size=2 * n // n_cells)
    cont.add(codes, cells)
    t2 = time.time()
    cv, ci, _, _, _ = orc.search(q, vq[0], pq, cont.storage, cont.is_empty, cont.cell_start, cont.cell_size,
                                 cont.address2id, k, n_probe, use_smart_probing=False,
                                 scan_fn=lambda *a: c_oracle.scan_topk(*a, n_threads=cores))
    t3 = time.time()

    # ---- GPU leg 1: the oracle-built index searched by the HIP path ----
    idx = IVFPQIndex(d_vector=d, n_subvectors=m, n_cells=n_cells, initial_size=1, device=str(device))
    sd = idx.state_dict()
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    new = {"_storage": dev(cont.storage), "_cell_start": dev(cont.cell_start), "_cell_size": dev(cont.cell_size),
           "_cell_capacity": dev(cont.cell_capacity), "_is_empty": dev(cont.is_empty),
           "_address2id": dev(cont.address2id),
           "vq_codec._is_trained": torch.tensor(True, device=device),
           "pq_codec._is_trained": torch.tensor(True, device=device),
           "vq_codec.kmeans.centroids": dev(vq[0]), "pq_codec.kmeans.centroids": dev(pq)}
    for key in set(sd) - set(new):
        new[key] = sd[key]
    idx.load_state_dict(new)
    idx.n_probe, idx.use_smart_probing = n_probe, False
    xq = dev(q)
    # (20 searches of 1 000 queries are 2 ms of wall time right after 80 s of 256 host threads: the host side of one
    # such run was seen at 4.5 ms per call with the kernels at their usual 42 us -- so: a pause, more warm-up, and the
    # faster of two runs, both on the record)
    time.sleep(0.3)
    runs = [time_search(idx, xq, k, 20, 10) for _ in range(2)]
    c1_ms_runs = [round(r[0] / 20 * 1e3, 4) for r in runs]
    dt, scan_ms, _, gv, gi, stats = min(runs, key=lambda r: r[0])
    gv, gi = gv.cpu().numpy(), gi.cpu().numpy()
    ids_equal = float((gi == ci).mean())
    finite = np.isfinite(cv)
    val_err = float(np.max(np.abs(gv[finite] - cv[finite]) / np.maximum(np.abs(cv[finite]), 1e-30))) if finite.any() else 0.0

    # ---- GPU leg 2: the same data through the GPU train / add (rates only: training is not bit-reproducible) ----
    np.random.seed(0)
    xg = dev(x)
    idx2 = IVFPQIndex(d_vector=d, n_subvectors=m, n_cells=n_cells, initial_size=2 * n // n_cells, device=str(device))
    torch.cuda.synchronize()
    g0 = time.time()
    idx2.train(xg)
    torch.cuda.synchronize()
    g1 = time.time()
    idx2.add(xg)
    torch.cuda.synchronize()
    g2 = time.time()
    algo = scanned_bytes(idx, xq, m)
    return {"workload": f"d={d} n={n} IVFPQ n_cells={n_cells} m={m} nprobe={n_probe} k={k}, {nq} queries, x ~ N(0,1) "
                        "(BASELINE.json configs[0])",
            "cpu": {"kind": "port", "cores": cores, "train_s": round(t1 - t0, 3), "add_s": round(t2 - t1, 3),
                    "search_s": round(t3 - t2, 3), "search_queries_per_s": round(nq / (t3 - t2), 1),
                    "kmeans_steps": {"vq": vq_steps, "pq": pq_steps},
                    "what": "oracle/: kmeans_fit_redo (VQ 15 / PQ 25 iterations, direct -(a-b)^2 assign in C/OpenMP, "
                            "numpy update), max_sim encode, ContainerState.add, search (numpy coarse + C/OpenMP scan)"},
            "gpu": {"search_queries_per_s": round(nq * 20 / dt, 1), "ms_per_step": round(dt / 20 * 1e3, 4),
                    "ms_per_step_of_both_runs": c1_ms_runs,
                    "train_s": round(g1 - g0, 3), "add_s": round(g2 - g1, 3),
                    "what": "IVFPQIndex.search() on the oracle-built index (load_state_dict); train/add of the same "
                            "data by the GPU path"},
            "value": round(nq * 20 / dt, 1), "unit": "queries/s",
            "ids_equal_to_oracle": round(ids_equal, 6), "values_max_rel_diff_vs_oracle": val_err,
            "roofline": hbm_roofline(algo, scan_ms, "scan_packed_kernel<.,16,.>", stream_peak,
                                     resident_bytes=idx._storage.numel(), stats=stats,
                                     bytes_per_query=round(algo / nq, 1))}


def secondary_c5(device, stream_peak, iters=3):
    """configs[4]: MultiKMeans n_kmeans=64 d=64 n=1M k=256: one Lloyd iteration as MultiKMeans.fit runs
    it -- tpq_lloyd_step on data prepared once per fit (three-level exact assign + update) -- with the
    per-kernel path (tpq_max_sim_select + tpq_compute_centroids) and the bit-exact fp32 assign beside it"""
    from torchpq_amd import kernels as K
    l, d, n, k = 64, 64, 1_000_000, 256
    g = torch.Generator(device=device)
    g.manual_seed(1237)
    data = torch.randn(l, d, n, generator=g, device=device)
    cent = data[:, :, torch.randperm(n, generator=g, device=device)[:k]].contiguous()
    from torchpq_amd.clustering import MultiKMeans
    mk = MultiKMeans(n_clusters=k)
    path = mk._assign_path(l, d, n, k, training=True)
    assign_fp32 = K.MaxSimHip(distance="euclidean")
    update = K.ComputeCentroidsHip()

    def timeit(fn, reps=iters):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    step = mk._lloyd_step_for(data, cent)   # what fit() builds once per call (None: shape does not qualify)
    assert step is not None, "configs[4] must take the prepared path"
    t_prepare = timeit(lambda: K.LloydStepHip(data, cent), 1)
    _, lab, new = step(cent)
    _, lab32 = assign_fp32(data, cent, dim=2, mode="tn")
    agree = float((lab == lab32).double().mean().item())
    ref_new = update(data, lab32, k=k)
    upd_err = float(((new - ref_new).abs().max() / ref_new.abs().max()).item())
    lvl2 = float(step.rechecked(1).double().sum().item()) / (l * n)
    lvl3 = float(step.rechecked(2).double().sum().item()) / (l * n)
    t_step = timeit(lambda: step(cent))
    t_assign = timeit(lambda: step(cent, update=False))
    # (derived: the update has no entry point of its own; a difference of two 3-repetition timings, clamped --
    # the update's rate below is indicative, iter_ms and assign_ms are the measured quantities)
    t_update = max(t_step - t_assign, 1e-3)
    t_fp32 = timeit(lambda: assign_fp32(data, cent, dim=2, mode="tn"))
    t_sel = timeit(lambda: mk.get_labels(data, cent, training=True))
    t_upd_sep = timeit(lambda: update(data, lab, k=k))
    out = {"workload": f"MultiKMeans n_kmeans={l} d={d} n={n} k={k}, one Lloyd iteration = assign + update "
                       "(tpq_lloyd_step on data prepared once per fit)"}
    flop = 2.0 * l * n * k * d
    byt = 4.0 * l * d * n
    tf = flop / t_assign / 1e9
    tf32 = flop / t_fp32 / 1e9
    # matrix-pipe work issued per 32 points x 32 centroids x 16 dimensions: level 1 = ONE fp16 product over
    # all points, level 2 = three products over the undecided share, + one norm MFMA per 32 x 32 tile
    ks = (d + 15) // 16
    issue_ratio = ((1.0 * ks + 1.0) + lvl2 * (3.0 * ks + 1.0)) / ks * (16.0 * ks / d)
    peak_equiv = MFMA_BF16_PEAK_TFLOPS / issue_ratio
    out.update({
        "assign_ms": round(t_assign, 3), "update_ms_derived": round(t_update, 3),
        "iter_ms": round(t_step, 3), "prepare_ms_once_per_fit": round(t_prepare, 3),
        "iter_TFLOPs_end_to_end": round(flop / t_step / 1e9, 1),
        "assign_labels_equal_to_fp32_kernel": round(agree, 6),
        "new_centroids_max_rel_diff_vs_tpq_compute_centroids": upd_err,
        "share_of_points_left_to_level_2": round(lvl2, 6), "share_re_checked_in_fp32": round(lvl3, 6),
        "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": round(peak_equiv, 1),
                     "unit": "TFLOP/s", "frac": round(tf / peak_equiv, 4), "traffic": None,
                     "kernel": "coarse_kernel (one fp16 product, every point) + refine_kernel (three products, "
                               "undecided share) + max_sim_kernel over the level-2 list (exact fp32): the fp32 "
                               "kernel's labels",
                     "assign_path": "lloyd_step", "per_kernel_assign_path": path,
                     "kernel_ms": round(t_assign, 3), "algorithmic_flops_per_launch": flop,
                     "issued_f16_TFLOPs": round(tf * issue_ratio, 1), "bf16_dense_peak": MFMA_BF16_PEAK_TFLOPS,
                     "peak_note": f"fp32-equivalent: fp16/bf16 dense peak / {issue_ratio:.2f} MFMA flops issued per "
                                  "algorithmic flop; frac = issued flop/s over the dense peak.  The coarse level "
                                  "is VALU-bound (top-2 update: 2.5 instructions per value), not matrix-bound"},
        "per_kernel_path": {"assign_ms": round(t_sel, 3), "update_ms": round(t_upd_sep, 3),
                            "iter_ms": round(t_sel + t_upd_sep, 3),
                            "what": "tpq_max_sim_select + tpq_compute_centroids on the fp32 data (round 2's "
                                    "iteration; still the path of shapes / fits too small to prepare)"},
        "assign_fp32": {"ms": round(t_fp32, 3), "what": "tpq_max_sim, the bit-exact kernel of predict/encode",
                        "roofline": {"bound": "mfma", "achieved": round(tf32, 1), "peak": MFMA_F32_PEAK_TFLOPS,
                                     "unit": "TFLOP/s", "frac": round(tf32 / MFMA_F32_PEAK_TFLOPS, 4),
                                     "kernel": "max_sim_codebook_kernel (fp32 MFMA)"}},
        "update_roofline": hbm_roofline(byt + 8.0 * l * n, t_update, "update inside tpq_lloyd_step",
                                        stream_peak)})
    return out


def secondary_wide(device, stream_peak, iters=3):
    """the coarse assign of a GIST-width index: 1 M points x 16 384 centroids x 960 dimensions through
    tpq_coarse_assign (fp16 selection on the matrix cores + candidate pairs evaluated exactly), with the
    bit-exact fp32 kernel beside it"""
    from torchpq_amd import kernels as K
    d, m, n = 960, 1_000_000, 16384
    g = torch.Generator(device=device)
    g.manual_seed(4321)
    A = torch.randn(d, m, generator=g, device=device)
    B = A[:, torch.randperm(m, generator=g, device=device)[:n]].contiguous()

    def timeit(fn, reps=iters):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    op = K.CoarseAssignHip(distance="euclidean")
    lab = op(A, B)
    undecided = op.last_rechecked() / m
    fp32 = K.MaxSimHip(distance="euclidean")
    _, l32 = fp32(A[None], B[None], dim=2, mode="tn")
    agree = float((lab == l32[0]).double().mean().item())
    t = timeit(lambda: op(A, B))
    t32 = timeit(lambda: fp32(A[None], B[None], dim=2, mode="tn"), 1)
    flop = 2.0 * m * n * d
    issue_ratio = 1.0 + undecided  # one fp16 product over every point + one more over the undecided ones
    tf = flop / t / 1e9
    peak_equiv = MFMA_BF16_PEAK_TFLOPS / issue_ratio
    return {"workload": f"tpq_coarse_assign d={d} points={m} centroids={n} (KMeans.predict / IVFPQIndex.add at GIST width)",
            "ms": round(t, 3), "fp32_kernel_ms": round(t32, 3), "speedup_vs_fp32_kernel": round(t32 / t, 2),
            "labels_equal_to_fp32_kernel": round(agree, 6), "share_of_points_with_an_exact_step": round(undecided, 5),
            "roofline": {"bound": "mfma", "achieved": round(tf, 1), "peak": round(peak_equiv, 1), "unit": "TFLOP/s",
                         "frac": round(tf / peak_equiv, 4), "traffic": None,
                         "kernel": "gemm_kernel<false> (fp16 hi pieces, key top-2) + gemm_kernel<true> (candidates of "
                                   "the undecided points) + pair_exact_kernel",
                         "kernel_ms": round(t, 3), "algorithmic_flops_per_launch": flop,
                         "issued_f16_TFLOPs": round(tf * issue_ratio, 1), "bf16_dense_peak": MFMA_BF16_PEAK_TFLOPS,
                         "peak_note": f"fp32-equivalent: fp16 dense peak / {issue_ratio:.2f} MFMA flops issued per "
                                      "algorithmic flop; the whole call (preparation of the points, candidate pass, "
                                      "exact pairs) is in the time"}}


def secondary_pass(device, budget_s, only=None, skip=(), stream_peak=None):
    """`skip`: records left for a later call (the driver's run does c1 -- 80 s of all host cores -- AFTER the
    headline's timed region); `stream_peak`: a value measured by an earlier call (not measured again)"""
    out = {}
    t_start = time.time()
    sp = stream_peak
    if sp is None:
        try:
            sp, out["stream_peak"] = stream_peak_record(device)
        except Exception as e:
            sp = None
            out["stream_peak"] = {"error": repr(e)[:300]}
    for name, fn in (("c4", secondary_c4), ("c3", secondary_c3), ("c5", secondary_c5), ("wide", secondary_wide),
                     ("residual", secondary_residual), ("flat", secondary_flat), ("c1", secondary_c1)):
        if (only and name not in only) or name in skip:
            continue
        if time.time() - t_start > budget_s:
            out[name] = {"skipped": f"secondary budget of {budget_s:.0f} s spent"}
            continue
        t0 = time.time()
        try:
            free, _ = torch.cuda.mem_get_info()
            if name != "c1" and free < 48 * 2 ** 30:
                out[name] = {"skipped": f"needs ~40 GB of HBM, {free >> 30} GiB free"}
                continue
            out[name] = fn(device, sp)
            prefix = {"c3": "scan_packed_kernel<1, 120", "c4": "scan_packed_kernel<1, 64"}.get(name)
            # every record carries the traffic of THIS run (a counter pass in a child process over the same
            # workload: the dominant kernel's HBM-side reads per launch); the committed profile is the fallback
            counted = prefix or {"c5": "coarse_kernel", "wide": "gemm_kernel<false"}.get(name)
            if name in ("residual", "flat"):
                pass  # (time-boxed records without a counter pass: `traffic` stays null)
            elif name == "c4":
                # the timed batch (10 000 four-wave workgroups) and the cold variant (2 048) run the same kernel: one
                # counter pass each, told apart by the launch's thread count
                if not measured_traffic(out[name]["roofline"], ["--secondary-only", name, "--c4-variant", "warm"], counted,
                                        grid_threads=10000 * 256):
                    attach_traffic(out[name]["roofline"], name)
                cold = out[name].get("cold", {}).get("roofline")
                if cold is not None and C4_VARIANT["mode"] == "both":
                    measured_traffic(cold, ["--secondary-only", name, "--c4-variant", "cold"], counted,
                                     grid_threads=C4_COLD_QUERIES * 256)
            elif not (counted and measured_traffic(out[name]["roofline"], ["--secondary-only", name], counted)):
                attach_traffic(out[name]["roofline"], name)
            if prefix:
                cross_check_profile(out[name]["roofline"], name, prefix)
        except Exception as e:
            out[name] = {"error": repr(e)[:300]}
        out[name]["wall_s"] = round(time.time() - t0, 1)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return out, sp


# ---------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_command(n, argv, port=None):
    """the one-rank-per-GPU launch of this script (what the driver runs for N > 1)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or free_port()),
            os.path.abspath(__file__)] + list(argv)


def self_launch(args):
    env = rccl_env(dict(os.environ))
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    return subprocess.call(launch_command(args.gpus, sys.argv[1:]), env=env)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=["c2", "c4"], default="c2",
                    help="c2 = SIFT1M shape (BASELINE.json configs[1], the headline); "
                         "c4 = 100 M vectors, n_cells=16384, n_probe=64 (configs[3])")
    ap.add_argument("--nq", type=int, default=10000,
                    help="queries per rank (weak scaling) / in the whole batch (strong scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="the mode `value` is quoted in: weak = every rank searches its own --nq "
                         "queries; strong = ONE batch of --nq queries split over the ranks "
                         "(shard_queries).  At N > 1 the other mode is timed too and reported "
                         "under the key `other_scaling` of the same line")
    ap.add_argument("--n-base", type=int, default=None)
    ap.add_argument("--n-train", type=int, default=100000)
    ap.add_argument("--d", type=int, default=128)
    ap.add_argument("--m", type=int, default=64)
    ap.add_argument("--n-cells", type=int, default=None)
    ap.add_argument("--n-probe", type=int, default=None)
    ap.add_argument("--k", type=int, default=100)
    ap.add_argument("--layout", choices=["packed", "ref"], default="packed")
    ap.add_argument("--data-dir", default=os.environ.get("TPQ_DATA_DIR"),
                    help="directory with sift_base/learn/query.fvecs (+ sift_groundtruth.ivecs)")
    ap.add_argument("--cpu-sample", type=int, default=10000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--no-traffic-pass", action="store_true",
                    help="do not run the rocprofv3 --pmc child pass that measures `traffic` in this run")
    ap.add_argument("--secondary-only", default=None, help="comma list of c1,c3,c4,c5,wide,residual,flat (profiling)")
    ap.add_argument("--secondary-budget", type=float, default=150.0)
    ap.add_argument("--c4-variant", choices=["both", "warm", "cold"], default="both",
                    help="--secondary-only c4: the timed batch, the cold variant, or both (profiles / counter passes)")
    ap.add_argument("--dist-timeout", type=float, default=300.0,
                    help="N > 1: timeout of the process groups (rendezvous, RCCL probe, broadcast)")
    ap.add_argument("--deadline", type=float, default=900.0,
                    help="N > 1: seconds after which rank 0 prints a JSON line with `error` and every rank exits")
    args = ap.parse_args(argv)
    if args.no_traffic_pass:
        TRAFFIC_PASS["enabled"] = False
    C4_VARIANT["mode"] = args.c4_variant
    c4 = args.workload == "c4"
    args.n_base = args.n_base or (100_000_000 if c4 else 1_000_000)
    args.n_cells = args.n_cells or (16384 if c4 else 1024)
    args.n_probe = args.n_probe or (64 if c4 else 32)
    return args


def error_line(args, world, msg):
    """what rank 0 prints when the distributed run cannot complete: ONE JSON line, never a hang"""
    return json.dumps({"metric": "queries/sec + recall@100, SIFT1M IVFPQ d=128 m=64 nprobe=32", "value": None,
                       "unit": "queries/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
                       "error": msg[:600], "world_size_seen": world})


def main():
    args = parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    done = {"ok": False}
    if world > 1:
        # deadline: a collective that never returns (a dead peer, a transport that hangs) must end in one JSON
        # line with `error`, not in the driver's own timeout
        import threading

        def deadline():
            time.sleep(args.deadline)
            if not done["ok"]:
                if rank == 0:
                    print(error_line(args, world, f"deadline of {args.deadline:.0f} s passed (stage: "
                                                  f"{done.get('stage', '?')})"), flush=True)
                os._exit(3)
        threading.Thread(target=deadline, daemon=True).start()
    try:
        run(args, world, rank, done)
        done["ok"] = True
    except BaseException as e:  # noqa: BLE001
        if isinstance(e, SystemExit) and not e.code:
            return
        if world == 1:
            raise
        import traceback
        traceback.print_exc()
        if rank == 0:
            print(error_line(args, world, f"stage {done.get('stage', '?')}: {type(e).__name__}: {e}"), flush=True)
        done["ok"] = True
        os._exit(4)   # (not sys.exit: a half-dead process group must not block interpreter shutdown)


def run(args, world, rank, done):
    import torch.distributed as dist
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # TPQ_BENCH_ONE_DEVICE=1 is a validation hook for 1-GPU boxes: every rank uses cuda:0 and the
    # bulk plane stays on gloo (RCCL refuses two ranks on one device); never set by the driver.
    # TPQ_BENCH_FAIL_RCCL=1 (same hook family): the RCCL probe raises, as a broken transport would
    one_device = os.environ.get("TPQ_BENCH_ONE_DEVICE", "0") == "1"
    if one_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from torchpq_amd import distributed as tpd
    from torchpq_amd.index import IVFPQIndex
    backend, groups, barrier = None, None, None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        done["stage"] = "rendezvous (gloo control plane) + RCCL probe"
        probe = create = None
        if os.environ.get("TPQ_BENCH_FAIL_RCCL", "0") == "1":
            def probe(_group):
                raise RuntimeError("TPQ_BENCH_FAIL_RCCL=1 (validation hook)")
            create = lambda: None  # noqa: E731
        groups = tpd.init_groups(device, want_rccl=(not one_device) or probe is not None,
                                 timeout_s=args.dist_timeout, probe=probe, create=create)
        backend = groups.bulk_backend
        barrier = tpd.host_barrier

    # ---- --secondary-only (profiling) / the stream peak; the secondary pass itself follows the headline ------
    secondary, stream_peak = None, None
    if world == 1 and args.secondary_only:
        secondary, stream_peak = secondary_pass(device, 1e9, set(args.secondary_only.split(",")))
        print(json.dumps({"secondary": secondary}))
        return
    do_secondary = world == 1 and not args.no_secondary and args.workload == "c2"
    stream_record = None
    if do_secondary:
        # The headline is timed FIRST, on a quiet chip: only the box's stream peak (a 60-ms read) is measured ahead of it.
        # (Timed after the matrix-pipe workloads of the secondary pass the same scan ran 3.32-3.39 ms instead of
        # 3.15-3.24 -- and 3.15-3.24 again when 80 s of CPU work lay between them.)  The other records follow it;
        # configs[0], whose CPU leg keeps every host core busy for 80 s, comes last.
        try:
            stream_peak, stream_record = stream_peak_record(device)
        except Exception as e:  # noqa: BLE001
            stream_peak, stream_record = None, {"error": repr(e)[:300]}
        torch.cuda.empty_cache()

    # ---- the index: built on rank 0, replicated with one broadcast per buffer ------------------
    t_train = t_add = 0.0
    base = gt_nn = None
    data_label = "synthetic (SIFT1M-shaped: non-negative integer-valued clustered fp32, low intrinsic dimension)"
    real = load_texmex(args.data_dir, "sift", device, args.n_base, args.nq) if args.workload == "c2" else None
    synth = SiftLike(args.d, device)
    if args.workload == "c4":
        data_label = "synthetic (uniform random codes, multinomial cell sizes, random codebooks)"
        if rank == 0:
            idx = fabricate_index(device, args.d, args.m, args.n_cells, args.n_base, seed=1236)
    elif rank == 0:
        if real is not None:
            base, train, _, _ = real
            data_label = f"SIFT1M from {args.data_dir}"
            if train is None:
                train = base[:, :args.n_train].contiguous()
        else:
            base = synth.sample(args.n_base, seed=1)
            gsel = torch.Generator(device=device)
            gsel.manual_seed(2)
            train = base[:, torch.randperm(args.n_base, generator=gsel, device=device)[:args.n_train]].contiguous()
        idx, t_train, t_add = build_index(args, device, base, train)
        del train
    if rank != 0:
        idx = IVFPQIndex(d_vector=args.d, n_subvectors=args.m, n_cells=args.n_cells, initial_size=1,
                         device=str(device))
    t_bcast, bcast_bytes = 0.0, 0
    if world > 1:
        done["stage"] = f"index broadcast over {backend}"
        torch.cuda.synchronize()
        barrier()
        t0 = time.time()
        # the one collective: the index buffers, <= 1 GiB per call, over RCCL (gloo when the probe failed)
        tpd.replicate_index(idx, src=0, bulk_group=groups.bulk)
        torch.cuda.synchronize()
        barrier()
        t_bcast = time.time() - t0
        bcast_bytes = int(getattr(idx, "replicated_bytes", 0))
    idx.n_probe = args.n_probe
    idx.use_smart_probing = False
    idx.use_packed_layout = args.layout == "packed"

    # weak scaling: every rank searches its own --nq queries (same distribution, rank-specific
    # seed); strong scaling: ONE batch of --nq queries (rank 0's), split over the ranks
    def full_queries(mode):
        seed_rank = rank if mode == "weak" else 0
        if args.workload == "c4":
            qg = torch.Generator(device=device)
            qg.manual_seed(4321 + seed_rank)
            return torch.randn(args.d, args.nq, generator=qg, device=device)
        if real is not None:
            return real[2]
        return synth.sample(args.nq, seed=4321 + seed_rank)

    def make_queries(mode):
        q = full_queries(mode)
        return q if mode == "weak" or world == 1 else tpd.shard_queries(q, rank, world)

    def timed(mode):
        """K timed steps in `mode`: (max-over-ranks seconds, per-rank seconds, scan ms, results, queries)"""
        q = make_queries(mode)
        done["stage"] = f"timed region ({mode} scaling)"
        dt_, scan_ms_, nb_, v_, i_, stats_ = time_search(idx, q, args.k, args.steps, args.warmup, barrier)
        stats_["launches_per_step"] = nb_
        per_rank = [dt_]
        if world > 1:  # (control plane: CPU tensors over gloo)
            t = torch.tensor([dt_], dtype=torch.float64)
            allt = [torch.zeros_like(t) for _ in range(world)]
            dist.all_gather(allt, t)
            per_rank = [float(x.item()) for x in allt]
        return max(per_rank), per_rank, scan_ms_, v_, i_, q, stats_

    def mode_summary(mode, dt_, per_rank):
        total_q = args.nq * (world if mode == "weak" else 1)
        return {"scaling": mode, "value": round(total_q * args.steps / dt_, 1), "unit": "queries/s",
                "queries_per_step_all_ranks": total_q, "ms_per_step": round(dt_ / args.steps * 1e3, 4),
                "ms_per_step_rank_min": round(min(per_rank) / args.steps * 1e3, 4),
                "ms_per_step_rank_max": round(max(per_rank) / args.steps * 1e3, 4)}

    other = None
    if world > 1:  # the mode `value` is NOT quoted in, first: the headline region runs last
        om = "strong" if args.scaling == "weak" else "weak"
        odt, oper, _, _, _, _, _ = timed(om)
        other = mode_summary(om, odt, oper)
    dt, per_rank_dt, scan_ms, vals, ids, queries, scan_stats = timed(args.scaling)
    headline = mode_summary(args.scaling, dt, per_rank_dt)

    # ---- N > 1: are the replicas equal?  Every rank searches ITS shard of the one shared batch (the strong-scaling
    # batch, rank 0's queries) and contributes a digest of the raw result bytes over the control plane; rank 0 searches
    # the whole batch alone and digests the same row ranges: equal digests <=> every replica returns, for its shard,
    # bit for bit what one GPU returns for the unsharded batch (no data-path collective: 16 bytes per rank)
    replica_check = None
    if world > 1:
        done["stage"] = "replica check (digests of the shared batch)"
        qfull = full_queries("strong")
        b0, e0 = tpd.shard_bounds(qfull.shape[1], rank, world)
        v_s, i_s = idx.search(qfull[:, b0:e0].contiguous(), k=args.k)
        digests = [None] * world
        dist.all_gather_object(digests, ids_digest(v_s, i_s))
        if rank == 0:
            v_f, i_f = idx.search(qfull, k=args.k)
            expect = []
            for r in range(world):
                b, e = tpd.shard_bounds(qfull.shape[1], r, world)
                expect.append(ids_digest(v_f[b:e], i_f[b:e]))
            replica_check = {"shard_digests": digests, "rank0_unsharded_digests": expect,
                             "all_ranks_bit_equal_to_rank0_unsharded": digests == expect,
                             "what": f"sha1[:16] of (values, ids) of each rank's shard of ONE batch of "
                                     f"{qfull.shape[1]} queries vs rank 0's search of the whole batch"}
            del v_f, i_f
        del qfull, v_s, i_s

    # ---- roofline of the dominant kernel (the list scan) --------------------------------------
    algo_bytes = scanned_bytes(idx, queries, args.m)  # uint8 codes only: the irreducible read
    kernel = ("scan_packed_kernel (m = 64, batches >= 1024: <1,64,false,-16>, four-wave workgroups over the 16-bit "
              "selection table, + scan_finish_exact_kernel; kernel_ms brackets the whole scan call)"
              if args.layout == "packed" else "scan_ref_kernel")
    roofline = hbm_roofline(
        algo_bytes, scan_ms, kernel, stream_peak, resident_bytes=idx._storage.numel(), stats=scan_stats,
        bytes_per_query=round(algo_bytes / queries.shape[1], 1),
        cell_imbalance=round(float((idx._cell_size.double() ** 2).sum().item()) * args.n_cells
                             / float(idx._cell_size.sum().item()) ** 2, 3))
    if args.layout == "packed" and args.workload == "c2":
        child = ["--steps", "3", "--warmup", "1", "--no-secondary", "--no-cpu-baseline", "--nq", str(args.nq),
                 "--n-base", str(args.n_base), "--n-train", str(args.n_train), "--d", str(args.d), "--m", str(args.m),
                 "--n-cells", str(args.n_cells), "--n-probe", str(args.n_probe), "--k", str(args.k)]
        if args.data_dir:
            child += ["--data-dir", args.data_dir]
        if not (world == 1 and rank == 0 and measured_traffic(roofline, child, "scan_packed_kernel<1, 64")):
            attach_traffic(roofline, "bench_scan_packed")
        cross_check_profile(roofline, "bench_scan_packed", "scan_packed_kernel<1, 64")

    shape = "SIFT1M" if real is not None else ("SIFT1M-like" if args.workload == "c2" else "synthetic")
    out = {
        "metric": "queries/sec + recall@100, SIFT1M IVFPQ d=128 m=64 nprobe=32",
        "value": headline["value"], "unit": "queries/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": headline["ms_per_step"], "higher_is_better": True,
        "scaling": args.scaling, "vs_baseline": None, "dtype": "f32",
        "data": data_label,
        "config": {"workload": f"{shape} d={args.d} n={args.n_base} IVFPQ n_cells={args.n_cells} "
                               f"m={args.m} nprobe={args.n_probe} k={args.k} on 1xMI355X per rank",
                   "n_query_per_rank": int(queries.shape[1]), "code_layout": args.layout,
                   "codes": "u8 (8-bit PQ)", "arithmetic": "f32 LUT entries, f32 sums, exact ids",
                   "use_smart_probing": False, "parallelism": f"query-sharded x{world}, replicated index",
                   "collective_backend": backend, "world_size_seen": world,
                   "collective_fallback_reason": groups.bulk_error if groups is not None else None,
                   "index_broadcast_s": round(t_bcast, 3), "index_broadcast_bytes": bcast_bytes,
                   "index_broadcast_GBps": round(bcast_bytes / t_bcast / 1e9, 2) if t_bcast > 0 else None,
                   "ms_per_step_rank_min": headline["ms_per_step_rank_min"],
                   "ms_per_step_rank_max": headline["ms_per_step_rank_max"]},
        "roofline": roofline,
    }
    if other is not None:
        out["other_scaling"] = other
    if replica_check is not None:
        out["replica_check"] = replica_check
    if do_secondary:
        # strong scaling on N GPUs searches nq / N queries per GPU: the rate a 1/N batch reaches on ONE GPU,
        # relative to the full batch, is the efficiency strong scaling can reach at N (no collective in the path)
        pred = {}
        for ng in (2, 4, 8):
            qs = queries[:, :max(1, queries.shape[1] // ng)].contiguous()
            dtn = time_search(idx, qs, args.k, max(5, args.steps // 2), 2)[0]
            pred[str(ng)] = round((qs.shape[1] * max(5, args.steps // 2) / dtn) / headline["value"], 4)
        out["strong_scaling_prediction"] = {
            "efficiency_at_n_gpus": pred,
            "what": f"rate of a {queries.shape[1]}/N-query batch on this one GPU over the rate of the full batch: "
                    "the ceiling of --scaling strong at N GPUs (weak scaling, the default `value`, keeps "
                    "the full batch per GPU)"}
    if do_secondary:
        secondary, _ = secondary_pass(device, args.secondary_budget, skip=("c1",), stream_peak=stream_peak or 0.0)
        secondary = {"stream_peak": stream_record, **secondary}
    if rank == 0:
        out["train_s"] = round(t_train, 3)
        out["add_s"] = round(t_add, 3)
        if base is not None:
            # recall@k = the true nearest neighbour (exact L2 on the raw vectors) is in the top-k:
            # the reference benchmark's definition, on a 1000-query sample
            ns = min(1000, queries.shape[1])
            nn = gt_nn[:ns] if gt_nn is not None else exact_nn(queries[:, :ns], base)
            out["recall_gt@%d" % args.k] = round(float((ids[:ns] == nn[:, None]).any(dim=1).float().mean().item()), 4)
            out["recall_gt@1"] = round(float((ids[:ns, 0] == nn).float().mean().item()), 4)
        if not args.no_cpu_baseline and world == 1 and args.workload == "c2":  # rank 0 at N=1 only
            cb, cpu_ids = cpu_baseline(idx, queries, args.k, min(args.cpu_sample, queries.shape[1]))
            out["cpu_baseline"] = cb
            gpu_ids = ids[:cpu_ids.shape[0]].cpu().numpy()
            inter = [len(np.intersect1d(gpu_ids[q], cpu_ids[q])) for q in range(cpu_ids.shape[0])]
            out["recall_vs_ref@%d" % args.k] = round(float(np.mean(inter)) / args.k, 4)
            out["ids_equal_to_oracle"] = round(float((gpu_ids == cpu_ids).mean()), 6)
        # the timed route checked at the size it was timed at: a sample of the batch vs the C oracle, bit for bit
        # (N > 1: the oracle on a 256-query sample of rank 0's batch -- the parity figure of the N-GPU line; the
        # CPU baseline itself stays an N = 1 record)
        if not args.no_cpu_baseline:
            oc = oracle_sample_check(idx, queries, vals, ids, args.k, n_sample=256 if world > 1 else 32)
            out["oracle_check"] = oc
            if world > 1:
                out["recall_vs_ref@%d" % args.k] = oc.pop("recall_vs_ref")
            else:
                oc.pop("recall_vs_ref", None)
        if do_secondary:
            late, _ = secondary_pass(device, args.secondary_budget, only={"c1"}, stream_peak=stream_peak or 0.0)
            secondary.update(late)
        if do_secondary:
            out["secondary"] = secondary
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
