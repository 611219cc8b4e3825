"""Single k-means problem (mirrors torchpq/clustering/KMeans.py:13-479): the l = 1 case of
MultiKMeans with 2-D tensors (data [d, n], centroids [d, n_clusters])."""

from ..CustomModule import CustomModule
from ..kernels import CoarseAssignHip
from .MultiKMeans import MultiKMeans


class KMeans(CustomModule):
    def __init__(self, n_clusters, n_redo=1, max_iter=100, tol=1e-4, distance="euclidean",
                 init_mode="random", verbose=0, sm_size=None, assign_precision="bf16x3"):
        super().__init__()
        self.verbose = verbose
        self.register_buffer("centroids", None)
        # not a registered child: the state_dict key must stay "centroids" (KMeans.py:75)
        object.__setattr__(self, "_multi", MultiKMeans(
            n_clusters, n_redo=n_redo, max_iter=max_iter, tol=tol, distance=distance,
            init_mode=init_mode, verbose=verbose, assign_precision=assign_precision))

    # knobs live on the batched engine
    n_clusters = property(lambda s: s._multi.n_clusters)
    distance = property(lambda s: s._multi.distance)
    init_mode = property(lambda s: s._multi.init_mode)

    def _knob(name):
        return property(lambda s: getattr(s._multi, name), lambda s, v: setattr(s._multi, name, v))

    max_iter = _knob("max_iter")
    n_redo = _knob("n_redo")
    tol = _knob("tol")
    assign_precision = _knob("assign_precision")
    del _knob

    calculate_error = staticmethod(MultiKMeans.calculate_error)
    calculate_inertia = staticmethod(MultiKMeans.calculate_inertia)

    remaining_memory = staticmethod(MultiKMeans.remaining_memory)
    does_it_fit = staticmethod(MultiKMeans.does_it_fit)

    def warmup_kernels(self):
        self._multi.warmup_kernels()

    @staticmethod
    def cos_sim(a, b, normalize=True, inplace=False):
        """[d, m] x [d, n] -> [m, n] (KMeans.py:155-181); never mutates its inputs"""
        return MultiKMeans.cos_sim(a[None], b[None], normalize=normalize)[0]

    @staticmethod
    def euc_sim(a, b, inplace=False):
        """[d, m] x [d, n] -> [m, n] negative squared L2 (KMeans.py:184-209)"""
        return MultiKMeans.euc_sim(a[None], b[None])[0]

    def kmeanspp(self, data):
        return self._multi.kmeanspp(data[None])[0]

    def sim(self, a, b, inplace=False, normalize=True):
        return self._multi.sim(a[None], b[None], normalize=normalize)[0]

    def initialize_centroids(self, data):
        return self._multi.initialize_centroids(data[None])[0]

    def get_labels(self, data, centroids):
        v, i = self._multi.get_labels(data[None], centroids[None])
        return v[0], i[0]

    def compute_centroids(self, data, labels):
        return self._multi.compute_centroids(data[None], labels[None])[0]

    def fit(self, data, centroids=None):
        assert data.is_contiguous(), "use .contiguous()"
        c0 = None if centroids is None else centroids[None].contiguous()
        labels = self._multi.fit(data[None], c0)
        self.register_buffer("centroids", self._multi.centroids[0].contiguous())
        return labels[0]

    # predict() = the coarse assign of IVFPQIndex.add: from this much work on (points x centroids x
    # dimensions) it runs tpq_coarse_assign -- error-bounded top-2 selection on the bf16 matrix
    # cores + exact re-check, the SAME labels as the fp32 kernel, 3-4x faster -- below it the three
    # launches cost more than they save
    fast_predict_min_work = 1 << 27

    def predict(self, query):
        assert self.centroids is not None, "kmeans is not trained"
        d, m = query.shape
        n = self.centroids.shape[1]
        # assign_precision="fp32" opts out of every selection kernel: tpq_max_sim everywhere
        if (self._multi.assign_precision != "fp32" and m * n * d >= self.fast_predict_min_work and n >= 64
                and CoarseAssignHip.supported(d, m, n)):
            centroids = self.centroids
            if self.distance == "cosine":  # normalised exactly as get_labels does, then inner product
                query = query / (query.norm(dim=-2, keepdim=True) + 1e-8)
                centroids = centroids / (centroids.norm(dim=-2, keepdim=True) + 1e-8)
            op = CoarseAssignHip(distance="euclidean" if self.distance == "euclidean" else "inner")
            return op(query, centroids)
        return self.get_labels(query, self.centroids)[1]

    def topk(self, query, k=128):
        assert self.centroids is not None, "kmeans is not trained"
        self._multi.centroids = self.centroids[None]
        v, i = self._multi.topk(query[None], k=k)
        return v[0], i[0]
