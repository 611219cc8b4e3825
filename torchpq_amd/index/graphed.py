"""HIP-graph replay of IVFPQIndex.search for fixed-shape, latency-bound serving batches.

A search() is a handful of small launches (coarse GEMM, select, scan, merge); for batches of
1-256 queries the host-side launch path (Python -> torch -> ctypes -> HIP) costs more than the GPU
work.  search() is sync-free, so the whole pipeline captures into one hipGraph and a replay is a
single launch.  (The reference has no equivalent: CuPy RawKernel launches are not capturable
through its wrappers.)
"""
import torch


class GraphedSearch:
    """``g = index.graphed_search(n_query, k); values, ids = g(x)``.

    The graph holds the index buffers' addresses: re-create it after add / remove / train /
    load_state_dict or a knob change (n_probe, use_smart_probing, ...).  The returned tensors are
    the graph's static outputs -- clone them if they must survive the next call."""

    def __init__(self, index, n_query, k, warmup=2):
        assert n_query >= 1
        self.index = index
        self.n_query = n_query
        self.k = k
        device = torch.device(index.device)
        self._codes_version = index._codes_version
        self.x = torch.zeros(index.d_vector, n_query, device=device, dtype=torch.float32)
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side):
            for _ in range(warmup):  # lazy state (scan-layout copy, part2 tables) is built here
                index.search(self.x, k=k)
        torch.cuda.current_stream(device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.values, self.ids = index.search(self.x, k=k)

    def __call__(self, x):
        assert x.shape == self.x.shape, f"graph was captured for queries of shape {tuple(self.x.shape)}"
        assert self._codes_version == self.index._codes_version, \
            "the index changed since the graph was captured: call index.graphed_search() again"
        self.x.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.values, self.ids
