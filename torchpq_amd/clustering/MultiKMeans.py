"""Batched Lloyd k-means on the GPU (mirrors torchpq/clustering/MultiKMeans.py:13-496).

Assign = tpq_max_sim (fp32 MFMA, bit-exact against the oracle) for predict / encode; inside fit()
the bf16 matrix cores where they apply (`_assign_path`): bounded selection + exact re-check with
the fp32 kernel's labels (tpq_max_sim_select, tpq_coarse_assign), or the 3-way split kernel
(tpq_max_sim_split, fp32-level accuracy); update = tpq_compute_centroids; the
Python below is only the Lloyd driver of the reference (fit :415-453, initialize_centroids :270-289).
"""
from time import time

import numpy as np
import torch

from .. import metric
from ..CustomModule import CustomModule
from ..kernels import CoarseAssignHip, ComputeCentroidsHip, LloydStepHip, MaxSimHip, MaxSimSelectHip


class MultiKMeans(CustomModule):
    """Run ``l`` independent k-means problems in parallel.

    data: [l, d_vector, n_data]; centroids: [l, d_vector, n_clusters]
    distance: 'euclidean' | 'cosine' | 'inner'
    """

    def __init__(self, n_clusters, n_redo=1, max_iter=100, tol=1e-4, distance="euclidean",
                 init_mode="random", verbose=0, sm_size=None, assign_precision="bf16x3"):
        super().__init__()
        assert assign_precision in ("fp32", "bf16x3")
        assert distance in ("euclidean", "cosine", "inner"), \
            "only euclidean / cosine / inner have a kernel branch (MultiKMeans.py:82-113)"
        assert init_mode in ("random", "kmeans++")
        self.n_redo = n_redo
        self.n_clusters = n_clusters
        self.max_iter = max_iter
        self.tol = tol
        self.verbose = verbose
        self.distance = distance
        self.init_mode = init_mode
        self.register_buffer("centroids", None)
        # "bf16x3" (default): the assign step may run on the bf16 matrix cores -- inside fit() per
        # `_assign_path`; in KMeans.predict / PQCodec.encode only through the selection kernels,
        # whose labels equal the fp32 kernel's bit for bit (error-bounded selection + exact
        # re-check).  "fp32" = tpq_max_sim, the bit-exact fp32-MFMA kernel, EVERYWHERE: fit(),
        # KMeans.predict (the coarse assign of IVFPQIndex.add) and PQCodec.encode.
        # MultiKMeans.predict() / get_labels() / kmeans++ always use the fp32 kernel.
        self.assign_precision = assign_precision
        self.max_sim_hip = MaxSimHip(dim=2, distance=distance)
        self.max_sim_split_hip = MaxSimHip(dim=2, distance=distance, precision="bf16x3")
        self.max_sim_select_hip = MaxSimSelectHip(distance="euclidean" if distance == "euclidean" else "inner")
        self.compute_centroids_hip = ComputeCentroidsHip()

    # -- memory helpers of the reference's public surface (:117-139); nothing here chunks by them:
    #    the assign kernel never materialises the [l, n, k] similarity tensor ----------------------
    @staticmethod
    def remaining_memory(device):
        """free bytes on `device` (HBM not reserved by the caching allocator)"""
        device = torch.device(device)
        if device.type != "cuda":
            raise RuntimeError("torchpq_amd runs on an AMD GPU (torch device 'cuda')")
        free, _ = torch.cuda.mem_get_info(device)
        return free + torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)

    @staticmethod
    def does_it_fit(size, device="cuda:0", dtype=torch.float):
        try:
            torch.empty(size, device=device, dtype=dtype)
        except Exception:
            return False
        return True

    def warmup_kernels(self):
        """the library is compiled ahead of time: nothing to warm up (the reference JIT-compiles
        its CuPy kernels on first use, :225-229)"""
        from .. import _lib
        _lib.load()

    # -- similarity helpers (reference: cos_sim :155-181, euc_sim :184-209, sim :211-223) ------
    @staticmethod
    def calculate_error(a, b):
        return (a - b).pow(2).sum()

    @staticmethod
    def calculate_inertia(a):
        return (-a).mean()

    @staticmethod
    def cos_sim(a, b, normalize=True, inplace=False):
        return metric.cosine_similarity(a, b, normalize=normalize)

    @staticmethod
    def euc_sim(a, b, inplace=False):
        return metric.negative_squared_l2_distance(a, b)

    def sim(self, a, b, inplace=False, normalize=True):
        """[l, d, m] x [l, d, n] -> [l, m, n]; never mutates its inputs (the reference's
        inplace=True path replaces data by |data|, KMeans.py:194-207)."""
        if self.distance == "euclidean":
            return self.euc_sim(a, b)
        if self.distance == "cosine":
            return self.cos_sim(a, b, normalize=normalize)
        return self.cos_sim(a, b, normalize=False)

    # -- Lloyd pieces ---------------------------------------------------------------------------
    def initialize_centroids(self, data):
        """random: the same np.random.choice index set for every sub-problem (:277-283)."""
        l, d, n = data.shape
        if self.init_mode == "random":
            index = np.random.choice(n, size=[self.n_clusters], replace=False)
            index = torch.from_numpy(index).to(data.device)
            centroids = data[:, :, index].clone()
        else:
            centroids = self.kmeanspp(data)
        return centroids

    def kmeanspp(self, data):
        """Farthest-point seeding as in the reference (:225-268): the next centroid is the point
        with the smallest maximum similarity to the centroids chosen so far."""
        l, d, n = data.shape
        centroids = torch.zeros(l, d, self.n_clusters, device=data.device, dtype=data.dtype)
        centroids[:, :, 0] = data[:, :, np.random.randint(n)]
        arange = torch.arange(l, device=data.device)
        for i in range(1, self.n_clusters):
            vals, _ = self.get_labels(data, centroids[:, :, :i].contiguous())
            index = vals.argmin(dim=-1)
            centroids[:, :, i] = data[arange, :, index]
        return centroids

    # below 12 dimensions the fp32 MFMA (K = 2 per instruction) wastes nothing and wins: the split
    # kernel pads every problem to K = 16 (6 + 1 bf16 MFMAs of 32 cycles against d/2 fp32 MFMAs of 64)
    split_min_d = 12

    def _assign_path(self, l, d, n, k, training):
        """which kernel the assign step runs on:
        "select"  codebook-sized problems (k <= 256, split_min_d <= d <= 64) inside fit():
                  tpq_max_sim_select -- bounded bf16 top-2 selection + exact re-check: the fp32
                  kernel's labels, bit for bit; maxima approximate (used for the inertia only);
        "coarse"  one problem with many centroids inside fit(): tpq_coarse_assign, same guarantees;
        "bf16x3"  other shapes with d <= 64 inside fit(): tpq_max_sim_split (fp32-level accuracy,
                  near-ties may resolve differently);
        "fp32"    everything else, and always outside fit(): the bit-exact kernel."""
        if not training or self.assign_precision != "bf16x3":
            return "fp32"
        if k <= 256 and d >= self.split_min_d and MaxSimSelectHip.supported(l, d, n, k):
            return "select"
        if (l == 1 and k >= 64 and n * k * d >= self.coarse_min_work and CoarseAssignHip.supported(d, n, k)):
            return "coarse"
        if d >= self.split_min_d and MaxSimHip.split_supported(d, n, k):
            return "bf16x3"
        return "fp32"

    # a single problem with many centroids (the coarse quantiser's training): the Lloyd loop takes its
    # labels from tpq_coarse_assign -- the fp32 kernel's labels, bit for bit, 3-4x faster; the
    # maxima that come with them are the selection's fast values (~1e-5 of the scale), used for the
    # inertia only
    coarse_min_work = 1 << 27

    def get_labels(self, data, centroids, training=False):
        """(max_sims [l, n], labels [l, n] int64); training=True: the Lloyd loop's assign, which may
        run on the bf16 matrix cores (`assign_precision`)"""
        if self.distance == "cosine":
            data = data / (data.norm(dim=-2, keepdim=True) + 1e-8)
            centroids = centroids / (centroids.norm(dim=-2, keepdim=True) + 1e-8)
        l, d, n = data.shape
        k = centroids.shape[2]
        path = self._assign_path(l, d, n, k, training)
        if path == "select":
            return self.max_sim_select_hip(data, centroids)
        if path == "coarse":
            op = CoarseAssignHip(distance="euclidean" if self.distance == "euclidean" else "inner")
            vals, labels = op(data[0], centroids[0], return_vals=True)
            return vals[None], labels[None]
        kernel = self.max_sim_split_hip if path == "bf16x3" else self.max_sim_hip
        return kernel(data, centroids, dim=2, mode="tn")

    def compute_centroids(self, data, labels):
        return self.compute_centroids_hip(data, labels, k=self.n_clusters)

    # Codebook-sized euclidean problems (the "select" shapes) with at least this much work per
    # iteration run the whole Lloyd iteration on PREPARED data (tpq_lloyd_prepare / tpq_lloyd_step):
    # the points are centred, scaled and split into fp16 pieces once per fit() -- they never change,
    # only the centroids do -- and every iteration is a three-level exact assign on those pieces + the
    # update.  Same labels as the fp32 kernel, bit for bit.  Preparing costs about two iterations and
    # a copy of the data in HBM; below `lloyd_min_iter` iterations it does not pay.
    lloyd_min_work = 1 << 31   # multiply-adds per iteration: l * n * k * d
    lloyd_min_iter = 4

    def _lloyd_step_for(self, data, centroids):
        """LloydStepHip for this fit(), or None when the shape / metric / size does not qualify (or HBM
        has no room for the prepared copy: the per-kernel path is then used, same results)"""
        l, d, n = data.shape
        k = centroids.shape[2]
        if (self.distance != "euclidean" or self.max_iter < self.lloyd_min_iter
                or self._assign_path(l, d, n, k, True) != "select" or l * n * k * d < self.lloyd_min_work
                or not LloydStepHip.supported(l, d, n, k)):
            return None
        try:
            return LloydStepHip(data, centroids)
        except torch.cuda.OutOfMemoryError:
            return None

    def fit(self, data, centroids=None):
        """Lloyd iterations; returns labels [l, n_data] of the best redo."""
        assert data.is_contiguous(), "use .contiguous()"
        best = None
        tm = time()
        step = False  # False: not decided yet; None: per-kernel path; else the prepared-data stepper
        for i in range(self.n_redo):
            if centroids is None:
                centroids = self.initialize_centroids(data)
            if step is False:  # the data is the same for every redo: prepared once
                step = self._lloyd_step_for(data, centroids)
            labels = maxsims = error = None
            for j in range(self.max_iter):
                if step is not None:
                    maxsims, labels, new_centroids = step(centroids)
                else:
                    maxsims, labels = self.get_labels(data, centroids, training=True)
                    new_centroids = self.compute_centroids(data, labels)
                error = self.calculate_error(centroids, new_centroids)
                centroids = new_centroids
                if self.verbose >= 3:
                    self.print_message(
                        f"----iteration {j} of {i}th redo, error={error.item()}, "
                        f"inertia={self.calculate_inertia(maxsims).item()}", 3)
                if error <= self.tol:  # one host sync per iteration, as in the reference (:437)
                    break
            inertia = self.calculate_inertia(maxsims)
            if best is None or inertia < best[0]:
                best = (inertia, centroids, labels)
            centroids = None
        self.register_buffer("centroids", best[1])
        del step  # the prepared copy of the data and the step workspace
        self.max_sim_select_hip.release()  # its l x n int32 lists
        self.print_message(
            f"finished {self.n_redo} redos in {round(time() - tm, 4)} sec, final_inertia: {best[0]}", 1)
        return best[2]

    def predict(self, query):
        assert self.centroids is not None, "kmeans is not trained"
        return self.get_labels(query, self.centroids)[1]

    def topk(self, query, k=128):
        """top-k closest centroids per query (:467-496): GEMM + row select."""
        assert self.centroids is not None, "kmeans is not trained"
        assert k <= self.n_clusters, "k is larger than number of clusters"
        if k == 1:
            v, i = self.get_labels(query, self.centroids)
            return v[..., None], i[..., None]
        from ..fn import Topk
        sims = self.sim(query, self.centroids)
        l, m, n = sims.shape
        v, i = Topk()(sims.reshape(l * m, n).contiguous(), k=k, dim=1)
        return v.reshape(l, m, k), i.reshape(l, m, k)
