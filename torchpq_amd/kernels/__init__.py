"""Kernel wrappers: torch tensors in, torch tensors out, HIP underneath.

Each class mirrors one CuPy RawKernel wrapper of the reference (torchpq/kernels/*Cuda.py):
same call signature and argument meaning, same asserts, outputs allocated with torch and
owned by the caller, launches asynchronous on the CURRENT torch stream, never synchronising.
There is no CPU path: tensors must live on the GPU.
"""
from __future__ import annotations

import torch

from .. import _lib
from .._lib import check, load, ptr, require_gpu, stream_ptr

__all__ = [
    "IVFPQTopkHip", "IVFPQTop1Hip", "ResidualPart1Hip", "ResidualSlotTermsHip", "AdcLutHip", "TopkSelectHip", "CoarseSelectHip", "CoarseProbeHip", "Top1SelectHip",
    "Top32SelectHip", "SmartProbingHip", "MaxSimHip", "ComputeCentroidsHip", "GetIOAHip",
    "GetWriteAddressHip", "GetCellByAddressHip", "GetIdByAddressHip", "GetAddressByIdHip", "GrowCellsHip", "PQDecodeHip",
    "ScatterCodesHip", "PackCodesHip", "packed_chunk_width", "PACKED_M",
]

# n_subvectors with an instantiated scan-layout kernel (= TPQ_PACKED_M_LIST in csrc/scan_device.h)
PACKED_M = (4, 8, 12, 16, 20, 24, 28, 32, 40, 48, 56, 64, 96, 120, 128)


def packed_chunk_width(m):
    return 16 if m % 16 == 0 else (8 if m % 8 == 0 else 4)


class IVFPQTopkHip:
    """IVF list scan + top-k.  Mirrors IVFPQTopkCuda (kernels/IVFPQTopkCuda.py:9-142);
    ``tpb``/``stack_capacity``/``sm_size`` are accepted for signature compatibility and
    ignored (the workgroup shape is fixed by the gfx950 kernel)."""

    def __init__(self, m=8, k=256, tpb=256, n_cs=4, stack_capacity=2, sm_size=None):
        assert k == 256  # 8-bit PQ only (IVFPQTopkCuda.py:21)
        assert n_cs == 4
        assert m % n_cs == 0
        self.m = m
        self.k = k
        self.tpb = tpb
        self.n_cs = n_cs
        self.n_cus = None
        # measurement hook (bench.py): when a list, every call appends a (start, stop) pair of
        # timing events recorded on the launch stream around the scan kernel(s)
        self.record_events = None
        self.last_n_split = None
        # tickets of the one-launch finish of split queries (tpq_ivfpq_*_tickets): caller-owned int32 [n_query],
        # zero between calls.  One buffer per (device, stream) -- calls that share a buffer must be ordered;
        # `ticket_buffer` overrides it (GraphedSearch hands in the buffer its graph owns).
        self.ticket_buffer = None
        self._ticket_cache = {}
        self.keep_workspace = False   # diagnostics: keep the last call's workspace in `last_workspace`
        self.last_workspace = None
        self.last_call = None         # diagnostics: the arguments of the last topk / topk_fused call (`last_route()`)

    def _tickets(self, n_query, n_split, device):
        """zeroed int32 [>= n_query] for this (device, current stream), or None (unsplit queries need none;
        inside a stream capture only a buffer handed in through `ticket_buffer` may be used: a fresh one
        would be zeroed by a captured memset on every replay)"""
        if n_split <= 1:
            return None
        if self.ticket_buffer is not None:
            t = self.ticket_buffer
            assert t.dtype == torch.int32 and t.numel() >= n_query and t.device == torch.device(device)
            return t
        if torch.cuda.is_current_stream_capturing():
            return None
        dev = torch.device(device)
        key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
        t = self._ticket_cache.get(key)
        if t is None or t.numel() < n_query:
            if len(self._ticket_cache) >= 64:
                self._ticket_cache.clear()
            t = torch.zeros(max(n_query, 1024), device=dev, dtype=torch.int32)
            self._ticket_cache[key] = t
        return t

    def _drop_tickets(self, device):
        """after a failed call the tickets may be left non-zero: never reuse them"""
        dev = torch.device(device)
        self._ticket_cache.pop((dev.index, torch.cuda.current_stream(dev).cuda_stream), None)

    def last_redone(self, n_query):
        """diagnostics (synchronises; needs keep_workspace): queries of the last packed scan that were redone exactly --
        by the one-launch finisher's own redo branch (ws_delta[q] == 1) or, on the routes that end with the flag-gated
        exact kernel (the large-batch route over the 16-bit table, the pools, the three-launch path: ws_delta holds a
        selection band there), by that kernel, which leaves kRedoneMark = -1 (csrc/scan_device.h)"""
        ws = self.last_workspace
        if ws is None:
            return None
        off = (n_query * 4 + 255) // 256 * 256
        d = ws[off:off + 4 * n_query].view(torch.float32)
        return int(((d == 1.0) | (d == -1.0)).sum().item())

    ROUTES = {0: "reference_layout", 1: "one_launch_finish", 2: "sorted_lists", 3: "pools", 8: "dump_f32",
              16: "dump_sel16", 17: "dump_sel16_w8", -1: "rejected"}

    def route(self, n_query, k, n_split=1, ds=0, n_probe=1, slots_hint=None, has_lut=True, packed=True,
              tickets=None, residual=False):
        """diagnostics: the kernels a call with these arguments runs (tpq_ivfpq_scan_route: the library's own rule,
        nothing is launched) -- one of ROUTES' names.  `tickets` defaults to what topk / topk_fused pass: the cached
        buffer of a split query outside a stream capture."""
        if tickets is None:
            tickets = n_split > 1
        code = load().tpq_ivfpq_scan_route(int(n_query), int(k), int(n_split), self.m, int(ds), int(n_probe),
                                            int(slots_hint or 0), int(bool(has_lut)),
                                            int(bool(packed) and self.m in PACKED_M), int(bool(tickets)),
                                            int(bool(residual)))
        return self.ROUTES.get(code, str(code))

    def last_route(self):
        """diagnostics: route(...) of the last topk / topk_fused call"""
        return None if self.last_call is None else self.route(**self.last_call)

    def _n_split(self, n_query, device, slots_hint=None):
        """Workgroups per query so that small batches still fill the chip (256 CUs x 2).
        ``slots_hint`` (expected slots scanned per query) caps the split so that every wave still
        walks >= 4 tiles: a wave that sees a single tile admits all 64 slots and the merge drowns."""
        if self.n_cus is None:
            self.n_cus = torch.cuda.get_device_properties(device).multi_processor_count
        # four 4-wave workgroups per CU for short codes (m <= 32), two 8-wave ones while the LUT is
        # <= 64 KiB, one 16-wave workgroup above (csrc/scan_device.h packed_waves)
        target = (4 if self.m <= 32 else 2 if self.m <= 64 else 1) * self.n_cus
        if n_query >= target:
            return 1
        split = max(1, min(64, target // max(n_query, 1)))
        if slots_hint is not None:
            waves = 4 if self.m <= 32 else 8 if self.m <= 64 else 16
            split = max(1, min(split, int(slots_hint) // (64 * waves * 4)))
        return split

    def topk(self, data, precomputed, is_empty, cell_start, cell_size, n_probe_list,
             n_candidates=None, packed=None, address2id=None, n_split=None, slots_hint=None):
        """
          data: [m // 4, n_data, 4] uint8           (CellContainer._storage)
          precomputed: [m, n_query, 256] float32    (PQCodec.precompute_adc)
          is_empty: [n_data] uint8, or None when no slot inside a cell is a tombstone
          cell_start / cell_size: [n_query, max_n_probe] int64
          n_probe_list: [n_query] int64
          n_candidates: k of the top-k (<= 1024)
          packed: optional scan-layout copy of `data` (enables the bank-conflict-free kernel)
          address2id: optional [n_data] int64; when given a third tensor (ids) is returned
        returns (values [n_query, k] descending, address [n_query, k][, ids])
        """
        n_data = data.shape[1]
        n_query, n_probe = cell_start.shape
        assert precomputed.shape == (self.m, n_query, self.k)
        assert data.shape[0] == self.m // self.n_cs
        assert data.shape[2] == self.n_cs
        assert cell_size.shape[1] == n_probe
        assert data.dtype == torch.uint8
        assert precomputed.dtype == torch.float32
        assert cell_start.dtype == cell_size.dtype == torch.int64
        assert n_probe_list.shape == (n_query,)
        assert n_probe_list.dtype == torch.int64
        if is_empty is not None:
            assert is_empty.shape[0] == n_data
            assert is_empty.dtype == torch.uint8
        if n_candidates is None:
            n_candidates = self.tpb
        assert 0 < n_candidates <= 1024
        require_gpu(data, precomputed, is_empty, cell_start, cell_size, n_probe_list, packed,
                    address2id)
        device = data.device
        k = n_candidates
        values = torch.empty(n_query, k, device=device, dtype=torch.float32)
        address = torch.empty(n_query, k, device=device, dtype=torch.int64)
        ids = torch.empty(n_query, k, device=device, dtype=torch.int64) if address2id is not None else None
        if n_query == 0:
            return (values, address) if ids is None else (values, address, ids)
        lib = load()
        if n_split is None:
            n_split = self._n_split(n_query, device, slots_hint)
        self.last_n_split = n_split  # diagnostics / tests: workgroups per query of the last call
        self.last_call = dict(n_query=n_query, k=k, n_split=n_split, ds=0, n_probe=n_probe, slots_hint=slots_hint,
                              has_lut=True, packed=packed is not None)
        ws_bytes = lib.tpq_ivfpq_scan_workspace_bytes(n_query, k, n_split, self.m)
        ws = torch.empty(max(ws_bytes, 1), device=device, dtype=torch.uint8) if ws_bytes else None
        ev = None
        if self.record_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(torch.cuda.current_stream(device))
        with torch.cuda.device(device):
            if packed is not None and self.m in PACKED_M:
                tickets = self._tickets(n_query, n_split, device)
                rc = lib.tpq_ivfpq_scan_topk_packed_tickets(
                    ptr(packed), ptr(data), ptr(precomputed), ptr(is_empty), ptr(cell_start),
                    ptr(cell_size), ptr(n_probe_list), ptr(values), ptr(address), ptr(address2id),
                    ptr(ids), n_data, n_query, n_probe, self.m, k, n_split, ptr(ws), ws_bytes,
                    ptr(tickets), int(slots_hint or 0), stream_ptr(device))
                if rc != 0:
                    self._drop_tickets(device)
                check(rc, "tpq_ivfpq_scan_topk_packed_tickets")
            else:
                rc = lib.tpq_ivfpq_scan_topk(
                    ptr(data), ptr(precomputed), ptr(is_empty), ptr(cell_start), ptr(cell_size),
                    ptr(n_probe_list), ptr(values), ptr(address), ptr(address2id), ptr(ids), n_data,
                    n_query, n_probe, self.m, k, n_split, ptr(ws), ws_bytes, stream_ptr(device))
                check(rc, "tpq_ivfpq_scan_topk")
        if ev is not None:
            ev[1].record(torch.cuda.current_stream(device))
            self.record_events.append(ev)
        if self.keep_workspace:
            self.last_workspace = ws
        return (values, address) if ids is None else (values, address, ids)

    def topk_fused(self, data, query, codebook, is_empty, cell_start, cell_size, n_probe_list,
                   n_candidates, distance="euclidean", packed=None, address2id=None, n_split=None,
                   slots_hint=None):
        """precompute_adc + topk in one pass: the LUT is built inside the scan workgroups
        (query [d, n_query] f32, codebook [m, ds, 256] f32); results identical to
        topk(precomputed=AdcLutHip()(query, codebook))."""
        n_data = data.shape[1]
        n_query, n_probe = cell_start.shape
        m, ds, kk = codebook.shape
        assert m == self.m and kk == self.k
        assert query.shape == (m * ds, n_query)
        assert query.dtype == codebook.dtype == torch.float32
        assert data.shape == (self.m // self.n_cs, n_data, self.n_cs) and data.dtype == torch.uint8
        assert cell_size.shape == (n_query, n_probe)
        assert cell_start.dtype == cell_size.dtype == torch.int64
        assert n_probe_list.shape == (n_query,) and n_probe_list.dtype == torch.int64
        assert 0 < n_candidates <= 1024
        query = query.contiguous()
        codebook = codebook.contiguous()
        require_gpu(data, query, codebook, is_empty, cell_start, cell_size, n_probe_list, packed,
                    address2id)
        device = data.device
        k = n_candidates
        values = torch.empty(n_query, k, device=device, dtype=torch.float32)
        address = torch.empty(n_query, k, device=device, dtype=torch.int64)
        ids = torch.empty(n_query, k, device=device, dtype=torch.int64) if address2id is not None else None
        if n_query == 0:
            return (values, address) if ids is None else (values, address, ids)
        lib = load()
        if n_split is None:
            n_split = self._n_split(n_query, device, slots_hint)
        self.last_n_split = n_split  # diagnostics / tests: workgroups per query of the last call
        self.last_call = dict(n_query=n_query, k=k, n_split=n_split, ds=ds, n_probe=n_probe, slots_hint=slots_hint,
                              has_lut=False, packed=packed is not None)
        ws_bytes = lib.tpq_ivfpq_scan_workspace_bytes(n_query, k, n_split, self.m)
        ws = torch.empty(max(ws_bytes, 1), device=device, dtype=torch.uint8)
        metric = _lib.METRIC_NEG_SQ_L2 if distance == "euclidean" else _lib.METRIC_INNER
        ev = None
        if self.record_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(torch.cuda.current_stream(device))
        with torch.cuda.device(device):
            tickets = self._tickets(n_query, n_split, device) if packed is not None else None
            rc = lib.tpq_ivfpq_search_fused_tickets(
                ptr(packed), ptr(data), ptr(query), ptr(codebook), ds, metric, ptr(is_empty),
                ptr(cell_start), ptr(cell_size), ptr(n_probe_list), ptr(values), ptr(address),
                ptr(address2id), ptr(ids), n_data, n_query, n_probe, self.m, k, n_split, ptr(ws),
                ws_bytes, ptr(tickets), int(slots_hint or 0), stream_ptr(device))
            if rc != 0:
                self._drop_tickets(device)
            check(rc, "tpq_ivfpq_search_fused_tickets")
        if self.keep_workspace:
            self.last_workspace = ws
        if ev is not None:
            ev[1].record(torch.cuda.current_stream(device))
            self.record_events.append(ev)
        return (values, address) if ids is None else (values, address, ids)

    # ---- residual PQ (pq_use_residual=True) ------------------------------------------------------
    def _residual(self, data, part1, part2, full, cells, base_sims, is_empty, cell_start, cell_size,
                  n_probe_list, n_candidates, address2id):
        n_data = data.shape[1]
        n_query, n_probe = cell_start.shape
        assert data.shape == (self.m // self.n_cs, n_data, self.n_cs)
        assert cell_size.shape == (n_query, n_probe)
        assert base_sims.shape == (n_query, n_probe)
        assert data.dtype == torch.uint8
        assert cell_start.dtype == cell_size.dtype == torch.int64
        assert base_sims.dtype == torch.float32
        assert n_probe_list.shape == (n_query,)
        assert n_probe_list.dtype == torch.int64
        if is_empty is not None:
            assert is_empty.shape == (n_data,) and is_empty.dtype == torch.uint8
        if n_candidates is None:
            n_candidates = self.tpb
        assert 0 < n_candidates <= 1024
        # logical [q][j][c] / [cell][j][c] / [q][p][j][c] order, whatever view the caller built
        part1 = None if part1 is None else part1.contiguous()
        part2 = None if part2 is None else part2.contiguous()
        full = None if full is None else full.contiguous()
        cells = None if cells is None else cells.contiguous()
        base_sims = base_sims.contiguous()
        require_gpu(data, part1, part2, full, cells, base_sims, is_empty, cell_start, cell_size,
                    n_probe_list, address2id)
        device = data.device
        k = n_candidates
        values = torch.empty(n_query, k, device=device, dtype=torch.float32)
        address = torch.empty(n_query, k, device=device, dtype=torch.int64)
        ids = torch.empty(n_query, k, device=device, dtype=torch.int64) if address2id is not None else None
        if n_query:
            with torch.cuda.device(device):
                check(load().tpq_ivfpq_scan_topk_residual(
                    ptr(data), ptr(part1), ptr(part2), ptr(full), ptr(cells), ptr(base_sims),
                    ptr(is_empty), ptr(cell_start), ptr(cell_size), ptr(n_probe_list), ptr(values),
                    ptr(address), ptr(address2id), ptr(ids), n_data, n_query, n_probe, self.m, k,
                    stream_ptr(device)), "tpq_ivfpq_scan_topk_residual")
        return (values, address) if ids is None else (values, address, ids)

    def topk_residual_packed(self, data, packed, part2, slot_term, cell_bound, cells, base_sims,
                             is_empty, cell_start, cell_size, n_probe_list, n_candidates,
                             part1=None, query=None, codebook=None, address2id=None, n_split=None,
                             slots_hint=None):
        """Residual scan on the scan layout (tpq_ivfpq_scan_topk_residual_packed): results equal
        topk_residual_precomputed bit for bit.  part1 [n_query, m, 256] or (query [d, n_query],
        codebook [m, ds, 256]) from which the workgroup builds it; part2 [n_cells, m, 256]
        contiguous; slot_term / cell_bound from ResidualSlotTermsHip."""
        n_data = data.shape[1]
        n_query, n_probe = cell_start.shape
        assert self.m in PACKED_M and packed is not None
        assert data.shape == (self.m // self.n_cs, n_data, self.n_cs) and data.dtype == torch.uint8
        assert part2.shape[1:] == (self.m, self.k) and part2.dtype == torch.float32
        assert part2.is_contiguous()
        assert slot_term.shape == (n_data,) and slot_term.dtype == torch.float32
        assert cell_bound.shape == (part2.shape[0],) and cell_bound.dtype == torch.float32
        assert cells.shape == cell_start.shape == cell_size.shape == base_sims.shape
        assert cells.dtype == cell_start.dtype == cell_size.dtype == torch.int64
        assert base_sims.dtype == torch.float32
        assert n_probe_list.shape == (n_query,) and n_probe_list.dtype == torch.int64
        assert 0 < n_candidates <= 1024
        ds = 0
        if part1 is not None:
            assert part1.shape == (n_query, self.m, self.k) and part1.dtype == torch.float32
            part1 = part1.contiguous()
        else:
            assert query is not None and codebook is not None
            ds = codebook.shape[1]
            assert codebook.shape == (self.m, ds, self.k) and query.shape == (self.m * ds, n_query)
            assert query.dtype == codebook.dtype == torch.float32
            query = query.contiguous()
            codebook = codebook.contiguous()
        cells = cells.contiguous()
        base_sims = base_sims.contiguous()
        require_gpu(data, packed, part1, query, codebook, part2, slot_term, cell_bound, cells,
                    base_sims, is_empty, cell_start, cell_size, n_probe_list, address2id)
        device = data.device
        k = n_candidates
        values = torch.empty(n_query, k, device=device, dtype=torch.float32)
        address = torch.empty(n_query, k, device=device, dtype=torch.int64)
        ids = torch.empty(n_query, k, device=device, dtype=torch.int64) if address2id is not None else None
        if n_query == 0:
            return (values, address) if ids is None else (values, address, ids)
        lib = load()
        if n_split is None:
            n_split = self._n_split(n_query, device, slots_hint)
        self.last_n_split = n_split  # diagnostics / tests: workgroups per query of the last call
        ws_bytes = lib.tpq_ivfpq_scan_workspace_bytes(n_query, k, n_split, self.m)
        ws = torch.empty(max(ws_bytes, 1), device=device, dtype=torch.uint8)
        ev = None
        if self.record_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record(torch.cuda.current_stream(device))
        with torch.cuda.device(device):
            check(lib.tpq_ivfpq_scan_topk_residual_packed(
                ptr(packed), ptr(data), ptr(part1), ptr(query), ptr(codebook), ds, ptr(part2),
                ptr(slot_term), ptr(cell_bound), ptr(cells), ptr(base_sims), ptr(is_empty),
                ptr(cell_start), ptr(cell_size), ptr(n_probe_list), ptr(values), ptr(address),
                ptr(address2id), ptr(ids), n_data, n_query, n_probe, self.m, k, n_split, ptr(ws),
                ws_bytes, stream_ptr(device)), "tpq_ivfpq_scan_topk_residual_packed")
        if self.keep_workspace:
            self.last_workspace = ws
        self.last_call = dict(n_query=n_query, k=k, n_split=n_split, ds=ds, n_probe=n_probe, slots_hint=slots_hint,
                              has_lut=part1 is not None, packed=True, residual=True)
        if ev is not None:
            ev[1].record(torch.cuda.current_stream(device))
            self.record_events.append(ev)
        return (values, address) if ids is None else (values, address, ids)

    def topk_residual(self, data, precomputed, base_sims, is_empty, cell_start, cell_size,
                      n_probe_list, n_candidates=None, address2id=None):
        """precomputed: [n_query, max_n_probe, m, 256] f32 -- one LUT per (query, probe)
        (kernels/IVFPQTopkCuda.py:144-210)."""
        n_query, n_probe = cell_start.shape
        assert precomputed.shape == (n_query, n_probe, self.m, self.k)
        assert precomputed.dtype == torch.float32
        return self._residual(data, None, None, precomputed, None, base_sims, is_empty, cell_start,
                              cell_size, n_probe_list, n_candidates, address2id)

    def topk_residual_precomputed(self, data, part1, part2, cells, base_sims, is_empty, cell_start,
                                  cell_size, n_probe_list, n_candidates=None, address2id=None):
        """part1 [n_query, m, 256], part2 [n_cells, m, 256] f32, cells [n_query, max_n_probe] int64
        (kernels/IVFPQTopkCuda.py:212-283)."""
        n_query = cell_start.shape[0]
        assert part1.shape == (n_query, self.m, self.k) and part2.shape[1:] == (self.m, self.k)
        assert part1.dtype == part2.dtype == torch.float32
        assert cells.shape == cell_start.shape and cells.dtype == torch.int64
        return self._residual(data, part1, part2, None, cells, base_sims, is_empty, cell_start,
                              cell_size, n_probe_list, n_candidates, address2id)


class ResidualSlotTermsHip:
    """Per-slot / per-cell constants of the packed residual scan (tpq_ivfpq_residual_slot_terms):
    slot_term [n_data] f32 = sum_j part2[cell(s), j, code_j(s)], cell_bound [n_cells] f32."""

    def __call__(self, data, part2, cell_start, cell_size):
        n_cells, m, kk = part2.shape
        n_data = data.shape[1]
        assert kk == 256 and data.shape == (m // 4, n_data, 4) and data.dtype == torch.uint8
        assert part2.dtype == torch.float32 and part2.is_contiguous()
        assert cell_start.shape == cell_size.shape == (n_cells,)
        assert cell_start.dtype == cell_size.dtype == torch.int64
        require_gpu(data, part2, cell_start, cell_size)
        slot_term = torch.empty(n_data, device=data.device, dtype=torch.float32)
        cell_bound = torch.empty(n_cells, device=data.device, dtype=torch.float32)
        with torch.cuda.device(data.device):
            check(load().tpq_ivfpq_residual_slot_terms(
                ptr(data), ptr(part2), ptr(cell_start), ptr(cell_size), ptr(slot_term),
                ptr(cell_bound), n_data, n_cells, m, stream_ptr(data.device)),
                "tpq_ivfpq_residual_slot_terms")
        return slot_term, cell_bound


class ResidualPart1Hip:
    """part1[q, j, c] = 2 * q_j . r_jc (index/IVFPQIndex.py:366-379), [n_query, m, 256] f32."""

    def __call__(self, query, codebook):
        m, ds, k = codebook.shape
        assert k == 256 and query.shape[0] == m * ds
        query = query.contiguous()
        codebook = codebook.contiguous()
        require_gpu(query, codebook)
        nq = query.shape[1]
        out = torch.empty(nq, m, 256, device=query.device, dtype=torch.float32)
        with torch.cuda.device(query.device):
            check(load().tpq_residual_part1(ptr(query), ptr(codebook), ptr(out), m, ds, nq,
                                            stream_ptr(query.device)), "tpq_residual_part1")
        return out


class IVFPQTop1Hip(IVFPQTopkHip):
    """k = 1 variant (kernels/IVFPQTop1Cuda.py:86-140): same kernel family, list of one."""

    def topk(self, *args, n_candidates=1, **kwargs):
        return super().topk(*args, n_candidates=n_candidates, **kwargs)


class AdcLutHip:
    """PQCodec.precompute_adc on the fp32 matrix cores (codec/PQCodec.py:62-75)."""

    def __call__(self, query, codebook, distance="euclidean"):
        """query [d, n_query] f32, codebook [m, ds, 256] f32 -> [m, n_query, 256] f32"""
        m, ds, k = codebook.shape
        assert k == 256
        assert query.shape[0] == m * ds
        assert query.dtype == codebook.dtype == torch.float32
        query = query.contiguous()
        codebook = codebook.contiguous()
        require_gpu(query, codebook)
        nq = query.shape[1]
        lut = torch.empty(m, nq, 256, device=query.device, dtype=torch.float32)
        metric = _lib.METRIC_NEG_SQ_L2 if distance == "euclidean" else _lib.METRIC_INNER
        with torch.cuda.device(query.device):
            check(load().tpq_adc_lut(ptr(query), ptr(codebook), ptr(lut), m, ds, nq, metric,
                                     stream_ptr(query.device)), "tpq_adc_lut")
        return lut


class TopkSelectHip:
    """Row-wise top-k, values descending (kernels/TopkSelectCuda.py:52-84,
    Top32SelectCuda.py:60-112, Top1SelectCuda.py)."""

    def __init__(self, tpb=256, queue_capacity=4, buffer_size=4):
        self.tpb = tpb

    def __call__(self, x, k=1, dim=1):
        assert len(x.shape) == 2
        assert dim in (1, -1), "only support last dimention"
        assert x.dtype == torch.float32
        assert 1 <= k <= 1024 and k <= x.shape[1]
        x = x.contiguous()
        require_gpu(x)
        rows, cols = x.shape
        vals = torch.empty(rows, k, device=x.device, dtype=torch.float32)
        inds = torch.empty(rows, k, device=x.device, dtype=torch.int64)
        with torch.cuda.device(x.device):
            check(load().tpq_topk_select(ptr(x), ptr(vals), ptr(inds), rows, cols, k,
                                         stream_ptr(x.device)), "tpq_topk_select")
        return vals, inds


class CoarseSelectHip:
    """negative_squared_l2_distance epilogue + row top-k in one pass (metric.py:89-96 + fn/Topk.py):
    dots [n_query, n_cells] = x^T C, a2 [n_query], b2 [n_cells] -> (sims, cells) [n_query, k]."""

    def __call__(self, dots, a2, b2, k):
        assert dots.dtype == a2.dtype == b2.dtype == torch.float32 and len(dots.shape) == 2
        rows, cols = dots.shape
        assert a2.shape == (rows,) and b2.shape == (cols,)
        assert 1 <= k <= 1024 and k <= cols
        dots, a2, b2 = dots.contiguous(), a2.contiguous(), b2.contiguous()
        require_gpu(dots, a2, b2)
        vals = torch.empty(rows, k, device=dots.device, dtype=torch.float32)
        inds = torch.empty(rows, k, device=dots.device, dtype=torch.int64)
        with torch.cuda.device(dots.device):
            check(load().tpq_coarse_select(ptr(dots), ptr(a2), ptr(b2), ptr(vals), ptr(inds), rows,
                                           cols, k, stream_ptr(dots.device)), "tpq_coarse_select")
        return vals, inds


Top1SelectHip = TopkSelectHip
Top32SelectHip = TopkSelectHip


class CoarseProbeHip:
    """The coarse step of IVFPQIndex.search in one call (tpq_ivfpq_coarse_probe): sims on the fp32
    matrix cores, row top-n_probe, list extents of the chosen cells, per-query probe count."""

    ROUTES = {"auto": _lib.PROBE_ROUTE_AUTO, "fp32": _lib.PROBE_ROUTE_FP32, "fp16": _lib.PROBE_ROUTE_FP16}

    def __init__(self, route="auto"):
        """route: which arithmetic SELECTS ("auto": the library's thresholds; "fp32": the fp32-MFMA kernels;
        "fp16": the fp16 selection pass + exact candidates wherever the shape allows) -- the result is the
        same, bit for bit, on every route (tpq_ivfpq_coarse_probe_route)"""
        assert route in self.ROUTES
        self.route = route

    @staticmethod
    def prepare(centroids):
        """the centroid-only part of the fp16 selection pass (tpq_ivfpq_coarse_probe_prepare), or None when the
        shape has none: a uint8 tensor to pass as `prepared` for as long as `centroids` does not change"""
        d, n_cells = centroids.shape
        assert centroids.dtype == torch.float32
        centroids = centroids.contiguous()
        require_gpu(centroids)
        lib = load()
        nbytes = lib.tpq_ivfpq_coarse_probe_prepared_bytes(d, n_cells)
        if nbytes == 0:
            return None
        out = torch.empty(nbytes, device=centroids.device, dtype=torch.uint8)
        with torch.cuda.device(centroids.device):
            check(lib.tpq_ivfpq_coarse_probe_prepare(ptr(centroids), d, n_cells, ptr(out), nbytes,
                                                     stream_ptr(centroids.device)), "tpq_ivfpq_coarse_probe_prepare")
        return out

    def __call__(self, query, centroids, cell_start, cell_size, n_probe, smart_temperature=None, prepared=None):
        """query [d, n_query] f32, centroids [d, n_cells] f32, cell_start / cell_size [n_cells] i64
        -> (topk_sims [n_query, n_probe] f32, cells, cell_start, cell_size [n_query, n_probe] i64,
            n_probe_list [n_query] i64)"""
        d, nq = query.shape
        n_cells = centroids.shape[1]
        assert centroids.shape[0] == d and query.dtype == centroids.dtype == torch.float32
        assert cell_start.shape == cell_size.shape == (n_cells,)
        assert cell_start.dtype == cell_size.dtype == torch.int64
        assert 1 <= n_probe <= min(n_cells, 1024)
        query = query.contiguous()
        centroids = centroids.contiguous()
        require_gpu(query, centroids, cell_start, cell_size)
        dev = query.device
        sims = torch.empty(nq, n_probe, device=dev, dtype=torch.float32)
        cells = torch.empty(nq, n_probe, device=dev, dtype=torch.int64)
        cs = torch.empty(nq, n_probe, device=dev, dtype=torch.int64)
        sz = torch.empty(nq, n_probe, device=dev, dtype=torch.int64)
        npl = torch.empty(nq, device=dev, dtype=torch.int64)
        if nq == 0:
            return sims, cells, cs, sz, npl
        lib = load()
        route = self.ROUTES[self.route]
        ws_bytes = lib.tpq_ivfpq_coarse_probe_route_workspace_bytes(d, nq, n_cells, route)
        ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
        t = float(smart_temperature) if smart_temperature else 0.0
        with torch.cuda.device(dev):
            check(lib.tpq_ivfpq_coarse_probe_route(
                ptr(query), ptr(centroids), ptr(cell_start), ptr(cell_size), ptr(sims), ptr(cells),
                ptr(cs), ptr(sz), ptr(npl), d, nq, n_cells, n_probe, t, route, ptr(prepared), ptr(ws), ws_bytes,
                stream_ptr(dev)), "tpq_ivfpq_coarse_probe_route")
        return sims, cells, cs, sz, npl


class SmartProbingHip:
    """n_probe_list from the entropy of the coarse similarities (index/IVFPQIndex.py:499-512)."""

    def __call__(self, topk_sims, temperature=30.0):
        assert topk_sims.dtype == torch.float32 and len(topk_sims.shape) == 2
        topk_sims = topk_sims.contiguous()
        require_gpu(topk_sims)
        rows, n_probe = topk_sims.shape
        out = torch.empty(rows, device=topk_sims.device, dtype=torch.int64)
        with torch.cuda.device(topk_sims.device):
            check(load().tpq_smart_probing(ptr(topk_sims), ptr(out), rows, n_probe,
                                           float(temperature), stream_ptr(topk_sims.device)),
                  "tpq_smart_probing")
        return out


class MaxSimHip:
    """Batched arg-max similarity, mode "tn" (kernels/MaxSimCuda.py:296-340): A [l, d, m] or
    [d, m], B [l, d, n] or [d, n] -> (vals, inds) over the n columns of B.

    precision="fp32" (default): tpq_max_sim, ascending-k fp32 fma chains on the fp32 MFMA,
    bit-exact against the oracle -- the encode / predict path.
    precision="bf16x3": tpq_max_sim_split, exact 3-way bf16 split of both operands on the bf16
    matrix cores (fp32-level accuracy, different rounding points; near-ties may resolve
    differently) -- the Lloyd loop of MultiKMeans.fit; shapes it does not cover fall back to the
    fp32 kernel (`split_supported`)."""

    def __init__(self, dim=2, distance="euclidean", precision="fp32", **_):
        assert distance in ("euclidean", "inner", "cosine")
        assert precision in ("fp32", "bf16x3")
        self.distance = distance
        self.dim = dim
        self.precision = precision

    @staticmethod
    def split_supported(d, m, n):
        return bool(load().tpq_max_sim_split_supported(int(d), int(m), int(n)))

    def __call__(self, A, B, dim=1, mode="tn"):
        assert mode == "tn", "only the 'tn' layout ([.., d, m] x [.., d, n]) is on the IVFPQ path"
        assert len(A.shape) == len(B.shape)
        two_d = len(A.shape) == 2
        if two_d:
            A, B, dim = A[None], B[None], dim + 1
        assert len(A.shape) == 3
        assert dim == 2, "arg-max is taken over the columns of B (dim=2; dim=1 for 2-D inputs)"
        assert A.shape[0] == B.shape[0] and A.shape[1] == B.shape[1]
        assert A.dtype == B.dtype == torch.float32
        A = A.contiguous()
        B = B.contiguous()
        require_gpu(A, B)
        l, d, m = A.shape
        n = B.shape[2]
        vals = torch.empty(l, m, device=A.device, dtype=torch.float32)
        inds = torch.empty(l, m, device=A.device, dtype=torch.int64)
        metric = _lib.METRIC_NEG_SQ_L2 if self.distance == "euclidean" else _lib.METRIC_INNER
        lib = load()
        split = self.precision == "bf16x3" and self.split_supported(d, m, n)
        fn, name = (lib.tpq_max_sim_split, "tpq_max_sim_split") if split else (lib.tpq_max_sim, "tpq_max_sim")
        with torch.cuda.device(A.device):
            check(fn(ptr(A), ptr(B), ptr(vals), ptr(inds), l, d, m, n, metric, stream_ptr(A.device)), name)
        if two_d:
            vals, inds = vals[0], inds[0]
        return vals, inds


class CoarseAssignHip:
    """Labels of MaxSimHip (fp32), bit for bit, for ONE problem with many centroids -- the coarse
    assign of IVFPQIndex.add / VQCodec.encode (kernels/MaxSimCuda.py:296-340 as called from
    clustering/KMeans.py:440-452): A [d, m], B [d, n] -> labels [m] int64, d <= 1024.  Error-bounded
    top-2 selection on the matrix cores; the points it leaves undecided get the exact kernel's own
    value -- over all centroids, or (from 4 096 centroids on, and for d > 128) for each of their
    candidates, the 2-3 centroids within twice the bound of the best (tpq_coarse_assign)."""

    # "auto": the library's size thresholds pick the path; "cascade": the fp16 cascade for every shape it
    # supports (tpq_coarse_assign_route; the parity tests set this to drive the cascade over small shapes)
    default_route = "auto"

    def __init__(self, distance="euclidean", route=None, **_):
        assert distance in ("euclidean", "inner", "cosine")
        assert route in (None, "auto", "cascade")
        self.distance = distance
        self.route = route
        self._last = None

    @staticmethod
    def supported(d, m, n):
        return bool(load().tpq_coarse_assign_supported(int(d), int(m), int(n)))

    def __call__(self, A, B, return_vals=False):
        """labels [m]; return_vals=True: (vals, labels) with vals the maximum similarity, approximate
        (within the selection bound) except for re-checked points"""
        assert A.dim() == 2 and B.dim() == 2 and A.shape[0] == B.shape[0]
        assert A.dtype == B.dtype == torch.float32
        A = A.contiguous()
        B = B.contiguous()
        require_gpu(A, B)
        d, m = A.shape
        n = B.shape[1]
        lib = load()
        inds = torch.empty(m, device=A.device, dtype=torch.int64)
        if m == 0:  # (empty tensors have null data pointers)
            self._last = None
            return (torch.empty(0, device=A.device), inds) if return_vals else inds
        route = (_lib.ASSIGN_ROUTE_CASCADE if (self.route or self.default_route) == "cascade"
                 else _lib.ASSIGN_ROUTE_AUTO)
        ws_bytes = lib.tpq_coarse_assign_route_workspace_bytes(d, m, n, route)
        ws = torch.empty(max(ws_bytes, 1), device=A.device, dtype=torch.uint8)
        metric = _lib.METRIC_NEG_SQ_L2 if self.distance == "euclidean" else _lib.METRIC_INNER
        vals = torch.empty(m, device=A.device, dtype=torch.float32) if return_vals else None
        with torch.cuda.device(A.device):
            check(lib.tpq_coarse_assign_route(ptr(A), ptr(B), ptr(vals) if return_vals else None, ptr(inds), d, m,
                                              n, metric, route, ptr(ws), ws_bytes, stream_ptr(A.device)),
                  "tpq_coarse_assign_route")
        self._last = (ws, lib.tpq_coarse_assign_count_offset(d, m, n))
        return (vals, inds) if return_vals else inds

    def last_rechecked(self):
        """diagnostics (synchronises): points of the last call that took an exact step (the exact kernel, or
        exact values of their candidates)"""
        if self._last is None:
            return 0
        ws, off = self._last
        return int(ws[off:off + 4].view(torch.int32).item())


class MaxSimSelectHip:
    """(vals, labels) for l codebook-sized problems (n <= 256 centroids, d <= 64): A [l, d, m], B [l, d, n].
    Labels are MaxSimHip's (fp32), bit for bit; vals are the selection's fast maxima, exact only for
    re-checked points (tpq_max_sim_select: bounded bf16 top-2 selection + exact re-check)."""

    def __init__(self, distance="euclidean", **_):
        assert distance in ("euclidean", "inner", "cosine")
        self.distance = distance
        self._ws = None

    @staticmethod
    def supported(l, d, m, n):
        return bool(load().tpq_max_sim_select_supported(int(l), int(d), int(m), int(n)))

    def __call__(self, A, B):
        assert A.dim() == 3 and B.dim() == 3 and A.shape[:2] == B.shape[:2]
        assert A.dtype == B.dtype == torch.float32
        A = A.contiguous()
        B = B.contiguous()
        require_gpu(A, B)
        l, d, m = A.shape
        n = B.shape[2]
        lib = load()
        vals = torch.empty(l, m, device=A.device, dtype=torch.float32)
        inds = torch.empty(l, m, device=A.device, dtype=torch.int64)
        if m == 0:
            return vals, inds
        ws_bytes = lib.tpq_max_sim_select_workspace_bytes(l, d, m, n)
        if self._ws is None or self._ws.numel() < ws_bytes or self._ws.device != A.device:
            self._ws = None
            self._ws = torch.empty(max(ws_bytes, 1), device=A.device, dtype=torch.uint8)
        metric = _lib.METRIC_NEG_SQ_L2 if self.distance == "euclidean" else _lib.METRIC_INNER
        with torch.cuda.device(A.device):
            check(lib.tpq_max_sim_select(ptr(A), ptr(B), ptr(vals), ptr(inds), l, d, m, n, metric,
                                         ptr(self._ws), ws_bytes, stream_ptr(A.device)), "tpq_max_sim_select")
        return vals, inds

    def release(self):
        """drop the cached workspace (l x m int32 lists)"""
        self._ws = None


class LloydStepHip:
    """One Lloyd iteration of MultiKMeans.fit on prepared data (tpq_lloyd_prepare / tpq_lloyd_step):
    the get_labels -> compute_centroids pair of the reference's driver
    (torchpq/clustering/MultiKMeans.py:415-453) for codebook-sized euclidean problems.

        step = LloydStepHip(data, centroids0)       # once per fit: centre, scale, split, fragment order
        maxsims, labels, new_centroids = step(centroids)

    labels are MaxSimHip's (fp32), bit for bit; maxsims are the selection's fast maxima (exact for
    re-checked points); new_centroids are the means of the labelled points summed from the fp16 pieces (h + m, two
    ulps of fp32 per element): within ~2e-7 of the scale of ComputeCentroidsHip()(data, labels, k), not bit-equal --
    a fit() that takes this path (MultiKMeans.lloyd_min_work / lloyd_min_iter) and one that does not agree to that
    tolerance per iteration."""

    @staticmethod
    def supported(l, d, m, n):
        return bool(load().tpq_lloyd_supported(int(l), int(d), int(m), int(n)))

    def __init__(self, data, centroids0):
        assert data.dim() == 3 and centroids0.dim() == 3 and data.shape[:2] == centroids0.shape[:2]
        assert data.dtype == centroids0.dtype == torch.float32
        require_gpu(data, centroids0)
        self.data = data.contiguous()
        centroids0 = centroids0.contiguous()
        l, d, m = self.data.shape
        n = centroids0.shape[2]
        assert self.supported(l, d, m, n), "shape not supported by tpq_lloyd_step (d <= 64, n <= 256)"
        self.shape = (l, d, m, n)
        lib = load()
        nbytes = lib.tpq_lloyd_prepared_bytes(l, d, m)
        self.prepared = torch.empty(nbytes, device=data.device, dtype=torch.uint8)
        # the step workspace (two l x m int lists, the sums) is allocated HERE, with the prepared copy: a caller
        # that guards the construction against torch.cuda.OutOfMemoryError (MultiKMeans.fit) then never meets one
        # inside its Lloyd loop
        self._ws = torch.empty(max(lib.tpq_lloyd_step_workspace_bytes(l, d, m, n), 1), device=data.device,
                               dtype=torch.uint8)
        with torch.cuda.device(data.device):
            check(lib.tpq_lloyd_prepare(ptr(self.data), ptr(centroids0), ptr(self.prepared), nbytes, l, d, m, n,
                                        stream_ptr(data.device)), "tpq_lloyd_prepare")

    def __call__(self, centroids, update=True):
        l, d, m, n = self.shape
        assert tuple(centroids.shape) == (l, d, n) and centroids.dtype == torch.float32
        centroids = centroids.contiguous()
        require_gpu(centroids)
        dev = self.data.device
        lib = load()
        vals = torch.empty(l, m, device=dev, dtype=torch.float32)
        inds = torch.empty(l, m, device=dev, dtype=torch.int64)
        new = torch.empty(l, d, n, device=dev, dtype=torch.float32) if update else None
        ws_bytes = lib.tpq_lloyd_step_workspace_bytes(l, d, m, n)
        if self._ws is None or self._ws.numel() < ws_bytes:
            self._ws = None
            self._ws = torch.empty(max(ws_bytes, 1), device=dev, dtype=torch.uint8)
        with torch.cuda.device(dev):
            check(lib.tpq_lloyd_step(ptr(self.data), ptr(self.prepared), ptr(centroids), ptr(new), ptr(vals),
                                     ptr(inds), l, d, m, n, ptr(self._ws), ws_bytes, stream_ptr(dev)),
                  "tpq_lloyd_step")
        return vals, inds, new

    def rechecked(self, level=2):
        """points per sub-problem the last step left undecided after level 1 (coarse pass) or level 2
        (= sent to the exact fp32 re-check); int32 [l], diagnostics"""
        l, d, m, n = self.shape
        off = load().tpq_lloyd_step_count_offset(l, d, m, n, int(level))
        return self._ws[off:off + 4 * l].view(torch.int32).clone()


class ComputeCentroidsHip:
    """K-means update (kernels/ComputeCentroidsCuda.py:43-81): data [l, d, n], labels [l, n]
    -> centroids [l, d, k]; empty clusters -> 0."""

    def __init__(self, de=1, dk=None, sm_size=None, **_):
        pass

    def __call__(self, data, labels, k, centroids=None):
        l, d, n = data.shape
        assert labels.shape == (l, n)
        assert data.dtype == torch.float32 and labels.dtype == torch.int64
        data = data.contiguous()
        labels = labels.contiguous()
        require_gpu(data, labels)
        lib = load()
        out = torch.empty(l, d, k, device=data.device, dtype=torch.float32)
        ws_bytes = lib.tpq_compute_centroids_workspace_bytes(l, d, k)
        ws = torch.empty(ws_bytes, device=data.device, dtype=torch.uint8)
        with torch.cuda.device(data.device):
            check(lib.tpq_compute_centroids(ptr(data), ptr(labels), ptr(out), l, d, n, k, ptr(ws),
                                            ws_bytes, stream_ptr(data.device)),
                  "tpq_compute_centroids")
        return out


class GetIOAHip:
    """Index of appearance (kernels/GetIOACuda.py:36-63): ioa[i] = #{j < i: labels[j] == labels[i]}."""

    def __init__(self, tpb=256):
        pass

    def __call__(self, labels, unique_labels=None, n_cells=None):
        assert labels.dtype == torch.int64 and len(labels.shape) == 1
        labels = labels.contiguous()
        require_gpu(labels)
        n = labels.shape[0]
        ioa = torch.empty_like(labels)
        if n == 0:
            return ioa
        if n_cells is None:
            n_cells = 2 ** 31 - 2  # sort on all 31 key bits
        lib = load()
        ws_bytes = lib.tpq_get_ioa_workspace_bytes(n)
        ws = torch.empty(ws_bytes, device=labels.device, dtype=torch.uint8)
        with torch.cuda.device(labels.device):
            check(lib.tpq_get_ioa(ptr(labels), ptr(ioa), n, int(n_cells), ptr(ws), ws_bytes,
                                  stream_ptr(labels.device)), "tpq_get_ioa")
        return ioa


class GetWriteAddressHip:
    """The ioa-th empty slot of each label's cell (kernels/GetWriteAddressV2Cuda.py:36-66)."""

    def __init__(self, tpb=256):
        pass

    def __call__(self, is_empty, div_start, div_size, labels, ioa):
        assert div_start.shape == div_size.shape
        assert ioa.shape == labels.shape
        require_gpu(is_empty, div_start, div_size, labels, ioa)
        n_slots = is_empty.shape[0]
        n_labels = labels.shape[0]
        out = torch.empty_like(labels)
        with torch.cuda.device(labels.device):
            check(load().tpq_get_write_address(ptr(is_empty), ptr(div_start), ptr(div_size),
                                               ptr(labels), ptr(ioa), ptr(out), n_slots, n_labels,
                                               stream_ptr(labels.device)), "tpq_get_write_address")
        return out


class GetCellByAddressHip:
    """address -> cell (kernels/GetDivByAddressV2Cuda.py:38-67); ``div_end`` = start + capacity."""

    def __init__(self, ta=4, tpb=256):
        pass

    def __call__(self, address, div_start, div_end):
        assert div_start.shape[0] == div_end.shape[0]
        address = address.contiguous()
        cap = (div_end - div_start).contiguous()
        div_start = div_start.contiguous()
        require_gpu(address, div_start, cap)
        out = torch.empty_like(address)
        with torch.cuda.device(address.device):
            check(load().tpq_get_cell_by_address(ptr(address), ptr(div_start), ptr(cap), ptr(out),
                                                 address.shape[0], div_start.shape[0],
                                                 stream_ptr(address.device)),
                  "tpq_get_cell_by_address")
        return out


class GetIdByAddressHip:
    """address -> id gather with -1 for invalid addresses (container/BaseContainer.py:58-65)."""

    def __call__(self, address2id, address):
        shape = address.shape
        flat = address.contiguous().view(-1)
        require_gpu(address2id, flat)
        out = torch.empty_like(flat)
        with torch.cuda.device(flat.device):
            check(load().tpq_get_id_by_address(ptr(address2id), address2id.shape[0], ptr(flat),
                                               ptr(out), flat.shape[0], stream_ptr(flat.device)),
                  "tpq_get_id_by_address")
        return out.view(shape)


class GetAddressByIdHip:
    """id -> address by comparing every id with every stored id (kernels/GetAddressByIdCuda.py,
    kernels/cuda/get_address_by_id.cu:8-44): the use_inverse_id_mapping=False path of
    BaseContainer.get_address_by_id; smallest matching address, -1 when absent."""

    def __init__(self, tpb=256):
        pass

    def __call__(self, address2id, ids):
        assert address2id.dtype == ids.dtype == torch.int64
        ids = ids.contiguous()
        require_gpu(address2id, ids)
        out = torch.empty_like(ids)
        with torch.cuda.device(ids.device):
            check(load().tpq_get_address_by_id(ptr(address2id), address2id.shape[0], ptr(ids), ptr(out),
                                               ids.shape[0], stream_ptr(ids.device)),
                  "tpq_get_address_by_id")
        return out


class GrowCellsHip:
    """CellContainer.expand in one pass (container/CellContainer.py:249-311): every cell moves to
    its place in the larger layout, new tails initialised free.  Returns the three new buffers."""

    def __call__(self, storage, address2id, is_empty, old_start, old_capacity, new_start, new_capacity,
                 new_slots, out=None):
        """out = (storage, address2id, is_empty) buffers of the new size to fill (must not alias the
        inputs), or None to allocate them"""
        g, old_slots, cs = storage.shape
        assert cs == 4 and storage.dtype == torch.uint8
        require_gpu(storage, address2id, is_empty, old_start, old_capacity, new_start, new_capacity)
        dev = storage.device
        if out is None:
            new_storage = torch.empty(g, new_slots, 4, device=dev, dtype=torch.uint8)
            new_a2i = torch.empty(new_slots, device=dev, dtype=torch.int64)
            new_empty = torch.empty(new_slots, device=dev, dtype=torch.uint8)
        else:
            new_storage, new_a2i, new_empty = out
            assert new_storage.shape == (g, new_slots, 4) and new_storage.is_contiguous()
            assert new_a2i.shape == (new_slots,) and new_empty.shape == (new_slots,)
            assert new_storage.dtype == torch.uint8 and new_a2i.dtype == torch.int64 and new_empty.dtype == torch.uint8
            assert new_storage.data_ptr() != storage.data_ptr() and new_a2i.data_ptr() != address2id.data_ptr()
        with torch.cuda.device(dev):
            check(load().tpq_grow_cells(ptr(storage), ptr(address2id), ptr(is_empty), ptr(old_start),
                                        ptr(old_capacity), ptr(new_start), ptr(new_capacity),
                                        ptr(new_storage), ptr(new_a2i), ptr(new_empty), old_slots,
                                        new_slots, old_start.shape[0], g * 4, stream_ptr(dev)),
                  "tpq_grow_cells")
        return new_storage, new_a2i, new_empty


class PQDecodeHip:
    """codes -> reconstruction (kernels/PQDecodeCuda.py:43-65)."""

    def __init__(self, tm=2, td=8):
        pass

    def __call__(self, codebook, code):
        m, d, k = codebook.shape
        assert code.shape[0] == m and k == 256
        assert code.dtype == torch.uint8
        codebook = codebook.contiguous()
        code = code.contiguous()
        require_gpu(codebook, code)
        n = code.shape[1]
        out = torch.empty(m * d, n, device=codebook.device, dtype=torch.float32)
        with torch.cuda.device(codebook.device):
            check(load().tpq_pq_decode(ptr(codebook), ptr(code), ptr(out), m, d, n,
                                       stream_ptr(codebook.device)), "tpq_pq_decode")
        return out


class ScatterCodesHip:
    """codes [m, n] -> _storage [m/4, cap, 4] (and the scan-layout copy) at `address`
    (CellContainer.set_data_by_address, container/CellContainer.py:213-239)."""

    def __call__(self, codes, address, storage, packed=None):
        m, n = codes.shape
        assert storage.shape[0] * storage.shape[2] == m and storage.shape[2] == 4
        assert address.shape[0] == n and address.dtype == torch.int64
        codes = codes.contiguous()
        address = address.contiguous()
        require_gpu(codes, address, storage, packed)
        with torch.cuda.device(codes.device):
            check(load().tpq_scatter_codes(ptr(codes), ptr(address), ptr(storage), ptr(packed), m, n,
                                           storage.shape[1], stream_ptr(codes.device)),
                  "tpq_scatter_codes")


class PackCodesHip:
    """(Re)build the MI355X scan layout from _storage for slots [begin, end)."""

    def __call__(self, storage, packed=None, begin=0, end=None):
        g, cap, cs = storage.shape
        assert cs == 4 and storage.dtype == torch.uint8
        m = g * cs
        w = packed_chunk_width(m)
        if packed is None:
            packed = torch.empty(m // w, cap, w, device=storage.device, dtype=torch.uint8)
        assert packed.shape == (m // w, cap, w)
        require_gpu(storage, packed)
        end = cap if end is None else end
        with torch.cuda.device(storage.device):
            check(load().tpq_ivfpq_pack_codes(ptr(storage), ptr(packed), cap, m, begin, end,
                                              stream_ptr(storage.device)), "tpq_ivfpq_pack_codes")
        return packed
