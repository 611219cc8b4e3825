"""Product quantiser, 8 bits per sub-vector (mirrors torchpq/codec/PQCodec.py:8-130)."""
from ..clustering import MultiKMeans
from ..kernels import AdcLutHip, PQDecodeHip
from .BaseCodec import BaseCodec


class PQCodec(BaseCodec):
    def __init__(self, d_vector, n_subvectors=8, n_clusters=256, distance="euclidean", verbose=0):
        super().__init__()
        assert d_vector % n_subvectors == 0
        assert n_clusters == 256, "only 8-bit PQ is on the IVFPQ path (IVFPQTopkCuda.py:21)"
        self.n_subvectors = n_subvectors
        self.n_clusters = n_clusters
        self.d_vector = d_vector
        self.d_subvector = d_vector // n_subvectors
        self.distance = distance
        self.verbose = verbose
        self.kmeans = MultiKMeans(n_clusters=n_clusters, distance=distance, max_iter=25,
                                  verbose=verbose)
        self._decode_hip = PQDecodeHip()
        self._adc_lut_hip = AdcLutHip()

    @property
    def codebook(self):
        """[n_subvectors, d_subvector, 256] or None before training"""
        return self.kmeans.centroids if self.is_trained else None

    def train(self, x):
        """x [d_vector, n_data] f32"""
        d_vector, n_data = x.shape
        assert d_vector == self.d_vector
        y = self.kmeans.fit(x.reshape(self.n_subvectors, self.d_subvector, n_data))
        self._trained(True)
        return y

    def precompute_adc(self, query):
        """query [d_vector, n_query] -> LUT [n_subvectors, n_query, 256] f32 with
        sum_j LUT[j, q, code_j] = -|q - decode(code)|^2 (euclidean) or the dot product."""
        assert self.is_trained, "codec is not trained"
        assert query.shape[0] == self.d_vector
        return self._adc_lut_hip(query, self.codebook, self.distance)

    def encode(self, x):
        """x [d_vector, n_data] f32 -> codes [n_subvectors, n_data] uint8"""
        assert self.is_trained, "codec is not trained"
        d_vector, n_data = x.shape
        assert d_vector == self.d_vector
        x = x.reshape(self.n_subvectors, self.d_subvector, n_data)
        # wide sub-vectors (d_subvector >= 12: 768-d embeddings at m = 64, ...): the bounded selection
        # + exact re-check of tpq_max_sim_select -- the fp32 kernel's labels, bit for bit, ~2x faster;
        # narrow ones (SIFT's 2, GIST's 8) stay on the fp32 MFMA, whose K = 2 wastes nothing
        km = self.kmeans
        # (kmeans.assign_precision = "fp32" opts out: tpq_max_sim everywhere)
        if (km.assign_precision != "fp32" and self.distance in ("euclidean", "inner")
                and self.d_subvector >= km.split_min_d
                and n_data * 256 * self.d_vector >= self.select_min_work
                and km.max_sim_select_hip.supported(self.n_subvectors, self.d_subvector, n_data, 256)):
            _, labels = km.max_sim_select_hip(x.contiguous(), self.codebook)
            km.max_sim_select_hip.release()
        else:
            _, labels = km.get_labels(x, self.codebook)
        return labels.byte()

    select_min_work = 1 << 27  # multiply-adds below which the extra launches do not pay

    def decode(self, code):
        """codes [n_subvectors, n_data] uint8 -> [d_vector, n_data] f32"""
        assert self.is_trained, "codec is not trained"
        assert code.shape[0] == self.n_subvectors
        return self._decode_hip(self.codebook, code)
