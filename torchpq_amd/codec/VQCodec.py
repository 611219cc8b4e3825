"""Coarse (inverted-file) quantiser (mirrors torchpq/codec/VQCodec.py:7-57)."""
from ..clustering import KMeans
from .BaseCodec import BaseCodec


class VQCodec(BaseCodec):
    def __init__(self, *args, **kwargs):
        super().__init__()
        self.kmeans = KMeans(*args, **kwargs)

    @property
    def codebook(self):
        """[d_vector, n_clusters]"""
        return self.kmeans.centroids

    def train(self, data):
        """data [d_vector, n_data] f32 -> labels [n_data] int64"""
        labels = self.kmeans.fit(data)
        self._trained(True)
        return labels

    def encode(self, input):
        """[d_vector, n_data] f32 -> [n_data] int64 (nearest cell)"""
        assert self.is_trained, "codec is not trained"
        return self.kmeans.predict(input)

    def decode(self, code):
        """[n] int64 -> [d_vector, n] f32"""
        assert self.is_trained, "Codec is untrained"
        return self.codebook[:, code].clone()
