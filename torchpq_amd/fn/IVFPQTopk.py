"""List-scan dispatch (mirrors torchpq/fn/IVFPQTopk.py:4-104).  The reference keeps four CUDA
kernels (k = 1, <= 256, <= 512, <= 1024); the HIP kernel is templated on the number of
64-candidate registers per wave and picked inside the C ABI."""
from ..kernels import IVFPQTopkHip


class IVFPQTopk:
    def __init__(self, n_subvectors, contiguous_size=4, sm_size=None):
        self.n_subvectors = n_subvectors
        self.contiguous_size = contiguous_size
        self._scan = IVFPQTopkHip(m=n_subvectors, n_cs=contiguous_size)

    def topk(self, data, precomputed, cell_start, cell_size, is_empty, n_probe_list, k=256,
             packed=None, address2id=None, slots_hint=None):
        assert 0 < k <= 1024
        return self._scan.topk(data=data, precomputed=precomputed, is_empty=is_empty,
                               cell_start=cell_start, cell_size=cell_size,
                               n_probe_list=n_probe_list, n_candidates=k, packed=packed,
                               address2id=address2id, slots_hint=slots_hint)

    def topk_fused(self, data, query, codebook, cell_start, cell_size, is_empty, n_probe_list, k=256,
                   distance="euclidean", packed=None, address2id=None, slots_hint=None):
        """PQCodec.precompute_adc + topk fused: the [m, n_query, 256] table never touches HBM"""
        assert 0 < k <= 1024
        return self._scan.topk_fused(data=data, query=query, codebook=codebook, is_empty=is_empty,
                                     cell_start=cell_start, cell_size=cell_size,
                                     n_probe_list=n_probe_list, n_candidates=k, distance=distance,
                                     packed=packed, address2id=address2id, slots_hint=slots_hint)

    def topk_residual(self, data, precomputed, cell_start, cell_size, base_sims, is_empty,
                      n_probe_list, k=256, address2id=None):
        """one LUT per (query, probe): precomputed [n_query, n_probe, m, 256] (:106-161)"""
        assert 0 < k <= 1024
        return self._scan.topk_residual(data=data, precomputed=precomputed, base_sims=base_sims,
                                        is_empty=is_empty, cell_start=cell_start,
                                        cell_size=cell_size, n_probe_list=n_probe_list,
                                        n_candidates=k, address2id=address2id)

    def topk_residual_precomputed(self, data, part1, part2, cell_start, cell_size, cells, base_sims,
                                  is_empty, n_probe_list=None, k=256, address2id=None):
        """LUT of a probe = part1[query] + part2[cell] (:163-228)"""
        assert 0 < k <= 1024
        return self._scan.topk_residual_precomputed(
            data=data, part1=part1, part2=part2, cells=cells, base_sims=base_sims,
            is_empty=is_empty, cell_start=cell_start, cell_size=cell_size,
            n_probe_list=n_probe_list, n_candidates=k, address2id=address2id)
