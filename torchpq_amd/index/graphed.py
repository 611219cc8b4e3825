"""HIP-graph replay of IVFPQIndex.search for fixed-shape, latency-bound serving batches.

A search() is a handful of small launches (coarse GEMM, select, scan, merge); for batches of
1-256 queries the host-side launch path (Python -> torch -> ctypes -> HIP) costs more than the GPU
work.  search() is sync-free, so the whole pipeline captures into one hipGraph and a replay is a
single launch.  (The reference has no equivalent: CuPy RawKernel launches are not capturable
through its wrappers.)
"""
import torch

from ..util import tensor_version


class GraphedSearch:
    """``g = index.graphed_search(n_query, k); values, ids = g(x)``.

    The graph holds the index buffers' addresses and has the knobs baked in as kernel arguments:
    re-create it after add / remove / train / load_state_dict or a knob change (n_probe,
    use_smart_probing, ...).  __call__ checks exactly that -- it compares a snapshot of every
    knob and of the identity (object, address, version counter) of every buffer search() reads
    with the live index and refuses to replay a stale graph; the captured buffers are kept alive
    by the snapshot, so a replay can never read freed memory.  The returned tensors are the
    graph's static outputs -- clone them if they must survive the next call."""

    KNOBS = ("n_probe", "use_smart_probing", "_smart_probing_temperature", "use_packed_layout",
             "use_fused_lut", "use_fused_probe", "use_cublas", "_use_precomputed", "pq_use_residual",
             "distance", "max_query_batch", "_has_holes", "_codes_version")

    @staticmethod
    def _buffers(index):
        """the device buffers a search() reads (None where the index does not have one)"""
        slot = index._slot_terms
        return {
            "vq_codebook": index.vq_codec.codebook, "pq_codebook": index.pq_codec.codebook,
            "_storage": index._storage, "_packed": index._packed if index._packed_valid else None,
            "_is_empty": index._is_empty, "_address2id": index._address2id,
            "_cell_start": index._cell_start, "_cell_size": index._cell_size,
            "_part2_by_cell": index._part2_by_cell,
            "slot_term": None if slot is None else slot[1],
            "cell_bound": None if slot is None else slot[2],
        }

    @classmethod
    def _snapshot(cls, index):
        knobs = tuple(getattr(index, name) for name in cls.KNOBS)
        bufs = cls._buffers(index)
        ident = tuple((name, None if t is None else (t.data_ptr(), tuple(t.shape), tensor_version(t)))
                      for name, t in bufs.items())
        return knobs, ident, bufs

    def __init__(self, index, n_query, k, warmup=2):
        assert n_query >= 1
        self.index = index
        self.n_query = n_query
        self.k = k
        device = torch.device(index.device)
        self.x = torch.zeros(index.d_vector, n_query, device=device, dtype=torch.float32)
        # the tickets of the one-launch finish of split queries (csrc/scan.hip) belong to this graph: zeroed
        # once here, left zero by every replay, freed with the graph's owner
        self._tickets = torch.zeros(max(n_query, 1), device=device, dtype=torch.int32)
        scan = index._ivfpq_topk._scan
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        prev = scan.ticket_buffer
        scan.ticket_buffer = self._tickets
        try:
            with torch.cuda.stream(side):
                for _ in range(warmup):  # lazy state (scan-layout copy, part2 tables) is built here
                    index.search(self.x, k=k)
            torch.cuda.current_stream(device).wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.values, self.ids = index.search(self.x, k=k)
        finally:
            scan.ticket_buffer = prev
        # taken AFTER the capture: lazily built state is in place; `_held` pins the tensors
        self._knobs, self._ident, self._held = self._snapshot(index)

    def stale_reason(self):
        """None while the captured graph still describes the index, else what changed"""
        knobs, ident, _ = self._snapshot(self.index)
        for name, old, new in zip(self.KNOBS, self._knobs, knobs):
            if old != new:
                return f"{name} changed ({old!r} -> {new!r})"
        for (name, old), (_, new) in zip(self._ident, ident):
            if old != new:
                return f"buffer {name} was replaced or written"
        return None

    def __call__(self, x):
        assert x.shape == self.x.shape, f"graph was captured for queries of shape {tuple(self.x.shape)}"
        why = self.stale_reason()
        if why is not None:
            raise RuntimeError(f"the index changed since the graph was captured ({why}): "
                               "call index.graphed_search() again")
        self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.values, self.ids
