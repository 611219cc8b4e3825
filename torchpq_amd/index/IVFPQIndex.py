"""IVFPQIndex: drop-in for torchpq.index.IVFPQIndex (reference index/IVFPQIndex.py:12-524).

Same constructor, methods, properties, asserts, tensor layouts ([d_vector, n] fp32 in, ids
int64 out, values descending with (-inf, -1) padding) and state_dict keys.  Everything below
the Python API runs in hand-written HIP kernels for gfx950 (libtorchpq_amd.so); the only
library math is the coarse query x cell-centroid GEMM and the one-off residual tables
(rocBLAS via torch.matmul/bmm, as the reference uses cuBLAS).
"""
import torch

from .. import metric, util
from ..codec import PQCodec, VQCodec
from ..container import CellContainer
from ..fn import IVFPQTopk, Topk
from ..kernels import CoarseProbeHip, CoarseSelectHip, SmartProbingHip


class IVFPQIndex(CellContainer):
    # search keeps the scan-layout copy of the codes at every m that has a kernel instantiation:
    # long codes gain from bank-conflict-free look-ups (m >= 56: 1.2-1.7x), short and medium ones
    # from several slots per lane per iteration (r02 sweep, scan layout vs reference layout:
    # m=28 1.97 vs 2.25 ms, 32 1.98 vs 2.19, 40 2.37 vs 2.67, 48 2.65 vs 2.94 per 10 000 queries).
    # (m with packed_max_short_subvectors < m < packed_min_subvectors would use the reference layout)
    packed_min_subvectors = 0
    packed_max_short_subvectors = 24
    # the LUT is built inside the scan workgroups (no [m, nq, 256] table in HBM) up to this
    # sub-vector length; the workgroup then reads m * ds KiB of L2-resident codebook per query
    fused_lut_max_subvector = 4

    def __init__(self, d_vector, n_subvectors=8, n_cells=128, initial_size=None,
                 expand_step_size=128, expand_mode="double", distance="euclidean",
                 device="cuda:0", pq_use_residual=False, verbose=0):
        if torch.device(device).type == "cuda":
            assert torch.cuda.is_available(), "cuda is not available"
            assert n_subvectors <= util.max_subvectors()
        assert d_vector % n_subvectors == 0
        assert n_subvectors % 4 == 0, "codes are stored 4 sub-quantizers per word (contiguous_size=4)"
        assert distance in ("euclidean", "cosine"), \
            "euclidean and cosine work end to end (MultiKMeans.py:82-113)"
        super().__init__(code_size=n_subvectors, n_cells=n_cells, dtype="uint8", device=device,
                         initial_size=initial_size, expand_step_size=expand_step_size,
                         expand_mode=expand_mode, use_inverse_id_mapping=True, contiguous_size=4,
                         verbose=verbose)
        self.d_vector = d_vector
        self.n_subvectors = n_subvectors
        self.d_subvector = d_vector // n_subvectors
        self.distance = distance
        self.verbose = verbose
        self.pq_use_residual = pq_use_residual
        self.n_probe = 1
        # residual search keeps a [n_cells, m, 256] table when it fits 4 GiB (IVFPQIndex.py:52-55)
        self._use_precomputed = bool(pq_use_residual and
                                     (n_cells * 256 * n_subvectors * 4) <= 4 * 1024 ** 3)
        self._precomputed_part2 = None
        self._part2_by_cell = None       # [n_cells, m, 256] contiguous copy for the scan kernels
        self._slot_terms = None          # (codes version, slot_term [capacity], cell_bound [n_cells])
        self._use_cublas = True
        self._use_smart_probing = True
        self._smart_probing_temperature = 30.0
        self._use_tensor_core = False
        self._fp16_scale_mode = "a"
        self.use_packed_layout = True   # MI355X scan layout (bank-conflict-free LDS look-ups)
        self.use_fused_lut = True       # build the ADC LUT inside the scan workgroups (no HBM table)
        self.use_fused_probe = True     # coarse sims + select + list extents + probe count: one call
        self.max_query_batch = 32768    # bounds the [m, nq, 256] LUT (m=64: 2 GiB per batch)

        self.vq_codec = VQCodec(n_clusters=n_cells, n_redo=1, max_iter=15, tol=1e-4,
                                distance="euclidean", init_mode="random", verbose=verbose)
        self.pq_codec = PQCodec(d_vector=d_vector, n_subvectors=n_subvectors, n_clusters=256,
                                distance=distance, verbose=verbose)
        self._ivfpq_topk = IVFPQTopk(n_subvectors=n_subvectors, contiguous_size=self.contiguous_size)
        self._topk = Topk()
        self._smart_probing = SmartProbingHip()
        self._coarse_select = CoarseSelectHip()
        self._coarse_probe = CoarseProbeHip()
        self.to(device)

    # ---- knobs (reference :89-232) ---------------------------------------------------------------
    @property
    def use_cublas(self):
        return self._use_cublas

    @use_cublas.setter
    def use_cublas(self, value):
        assert type(value) is bool
        self._use_cublas = value

    @property
    def use_tensor_core(self):
        return self._use_tensor_core

    @use_tensor_core.setter
    def use_tensor_core(self, value):
        """True: the coarse step SELECTS on the fp16 matrix cores wherever the shape allows it (d <= 128) and
        gives every candidate cell the fp32 kernel's own similarity -- the cells, their order and the
        similarities are unchanged, bit for bit (the reference's knob, :98-125, trades accuracy for the speed).
        False (default): the library's thresholds decide (the same pass from 2 048 cells on)."""
        assert type(value) is bool
        assert self.use_cublas
        self._use_tensor_core = value
        self._coarse_probe.route = "fp16" if value else "auto"

    @property
    def fp16_scale_mode(self):
        return self._fp16_scale_mode

    @fp16_scale_mode.setter
    def fp16_scale_mode(self, value):
        assert value in ["a", "b", "both", "none"]
        self._fp16_scale_mode = value

    @property
    def use_smart_probing(self):
        return self._use_smart_probing

    @use_smart_probing.setter
    def use_smart_probing(self, value):
        assert type(value) is bool
        self._use_smart_probing = value

    @property
    def smart_probing_temperature(self):
        return self._smart_probing_temperature

    @smart_probing_temperature.setter
    def smart_probing_temperature(self, value):
        assert value > 0
        assert self.use_smart_probing, "set use_smart_probing to True first"
        self._smart_probing_temperature = value

    @property
    def use_precomputed(self):
        return self._use_precomputed

    @use_precomputed.setter
    def use_precomputed(self, value):
        assert type(value) is bool
        if value:
            assert self.pq_use_residual, " `use_precomputed=True` is only valid when `pq_use_residual` is True"
            assert (self.pq_codec.is_trained and self.vq_codec.is_trained), "index is not trained"
            self.precompute_part2()
        else:
            self._drop_part2()
        self._use_precomputed = value

    def _drop_part2(self):
        self._precomputed_part2 = None
        self._part2_by_cell = None
        self._slot_terms = None

    def precompute_part2(self):
        """[m, n_cells, 256]: -2 c_j . r_jc - |r_jc|^2 (reference :160-170; one-off library bmm)"""
        pq_codebook = self.pq_codec.codebook
        vq_codebook = self.vq_codec.codebook.reshape(self.n_subvectors, self.d_subvector, self.n_cells)
        part2 = (torch.bmm(vq_codebook.transpose(-1, -2), pq_codebook) * -2
                 - pq_codebook.norm(dim=1).pow(2)[:, None])
        # one resident copy, in the [n_cells, m, 256] order the scan kernels read;
        # `_precomputed_part2` keeps the reference's [m, n_cells, 256] shape as a view of it
        self._part2_by_cell = part2.transpose(0, 1).contiguous()
        del part2
        self._precomputed_part2 = self._part2_by_cell.transpose(0, 1)
        self._slot_terms = None

    def _residual_slot_terms(self):
        """(slot_term, cell_bound) of the packed residual scan, rebuilt when the codes changed"""
        if self._slot_terms is None or self._slot_terms[0] != self._codes_version:
            from ..kernels import ResidualSlotTermsHip
            st, cb = ResidualSlotTermsHip()(self._storage, self._part2_by_cell, self._cell_start,
                                            self._cell_size)
            self._slot_terms = (self._codes_version, st, cb)
        return self._slot_terms[1], self._slot_terms[2]

    def _codec_knob(codec, attr, typed):
        def getter(self):
            return getattr(getattr(self, codec).kmeans, attr)

        def setter(self, value):
            if typed:
                assert type(value) is int
                assert value > 0
            assert not getattr(self, codec).is_trained, f"{codec} is already trained"
            setattr(getattr(self, codec).kmeans, attr, value)
        return property(getter, setter)

    vq_codec_max_iter = _codec_knob("vq_codec", "max_iter", True)
    vq_codec_n_redo = _codec_knob("vq_codec", "n_redo", True)
    vq_codec_tolerance = _codec_knob("vq_codec", "tol", False)
    pq_codec_max_iter = _codec_knob("pq_codec", "max_iter", True)
    pq_codec_n_redo = _codec_knob("pq_codec", "n_redo", True)
    pq_codec_tolerance = _codec_knob("pq_codec", "tol", False)
    del _codec_knob

    def _after_load_state_dict(self):
        self.to(self.device)
        self._drop_part2()
        super()._after_load_state_dict()

    # ---- train / encode / add (reference :234-364) ------------------------------------------------
    def train(self, x, force_retrain=False):
        """x [d_vector, n_data] f32: coarse k-means (n_cells) then 256-centroid PQ codebooks."""
        if self.vq_codec.is_trained and self.pq_codec.is_trained and not force_retrain:
            self.print_message("index is already trained", 1)
            return
        assert len(x.shape) == 2
        assert x.shape[0] == self.d_vector
        x = x.to(self.device)
        if self.distance == "cosine":
            x = util.normalize(x, dim=0)
        x = x.contiguous()
        self.print_message("start training VQ codec...", 1)
        code = self.vq_codec.train(x)
        self.print_message("start training PQ codec...", 1)
        if self.pq_use_residual:
            # PQ is learnt on x - centroid(x); the caller's tensor is left untouched (the reference
            # subtracts and re-adds in place, :246-254)
            self.pq_codec.train((x - self.vq_codec.decode(code)).contiguous())
            self._drop_part2()
        else:
            self.pq_codec.train(x)
        self.print_message("index is trained successfully!", 1)

    def encode(self, x):
        """x [d_vector, n] f32 -> PQ codes [n_subvectors, n] uint8"""
        assert len(x.shape) == 2
        assert x.shape[0] == self.d_vector
        x = x.to(self.device)
        if self.distance == "cosine":
            x = util.normalize(x)
        x = x.contiguous()
        if self.pq_use_residual:  # (pq_code, vq_code) of the residual x - centroid (:276-281)
            return self._encode_raw(x)
        return self.pq_codec.encode(x)

    def decode(self, x):
        """codes [n_subvectors, n] uint8 -> [d_vector, n] f32; with pq_use_residual `x` is the
        pair (pq_code, vq_code) and the result is centroid + decoded residual (:301-309)"""
        if self.pq_use_residual:
            assert len(x) == 2
            pq_code, vq_code = x
            assert pq_code.shape[0] == self.n_subvectors
            assert pq_code.shape[1] == vq_code.shape[0]
            return self.vq_codec.decode(vq_code.to(self.device)) + self.pq_codec.decode(pq_code.to(self.device))
        assert len(x.shape) == 2
        assert x.shape[0] == self.n_subvectors
        return self.pq_codec.decode(x.to(self.device))

    def add(self, x, ids=None, return_address=False):
        """x [d_vector, n] f32, optional ids [n] int64 (default arange + max_id + 1);
        returns ids (and the slot addresses if return_address)."""
        assert len(x.shape) == 2
        assert x.shape[0] == self.d_vector
        x = x.to(self.device)
        if self.distance == "cosine":
            x = util.normalize(x)
        x = x.contiguous()
        if self.pq_use_residual:
            codes, assigned_cells = self._encode_raw(x)
        else:
            assigned_cells = self.vq_codec.encode(x)
            codes = self.pq_codec.encode(x)
        return super().add(codes, cells=assigned_cells, ids=ids, return_address=return_address)

    def _encode_raw(self, x):
        """residual encode of an already normalised x"""
        vq_code = self.vq_codec.encode(x)
        return self.pq_codec.encode((x - self.vq_codec.decode(vq_code)).contiguous()), vq_code

    # ---- residual tables (reference :366-405) -----------------------------------------------------
    def precomputed_adc_residual_precomputed(self, x):
        """(part1 [n_query, m, 256] = 2 q_j.r_jc,  part2 [n_cells, m, 256])"""
        from ..kernels import ResidualPart1Hip
        part1 = ResidualPart1Hip()(x, self.pq_codec.codebook)
        if self._precomputed_part2 is None:
            self.precompute_part2()
        return part1, self._part2_by_cell

    def precomputed_adc_residual(self, x, cells):
        """[n_query, n_probe, m, 256]: one LUT per (query, probe); memory-hungry, used only when
        the part2 table is disabled (use_precomputed=False)"""
        n_query, n_probe = cells.shape
        pq_codebook = self.pq_codec.codebook
        xs = x.reshape(self.n_subvectors, self.d_subvector, n_query).transpose(-1, -2)
        part1 = 2 * (xs @ pq_codebook).permute(1, 0, 2) - pq_codebook.norm(dim=1).pow(2)[None]
        vq = self.vq_codec.codebook.reshape(self.n_subvectors, self.d_subvector, self.n_cells)
        vq = vq[:, :, cells].permute(0, 2, 3, 1)                       # [m, nq, n_probe, ds]
        part2 = -2 * (vq @ pq_codebook[:, None].expand(-1, n_query, -1, -1))  # [m, nq, n_probe, 256]
        return (part1[:, None] + part2.permute(1, 2, 0, 3)).contiguous()

    # ---- search (reference :407-524) ---------------------------------------------------------------
    def search_cells(self, x, cells, base_sims=None, n_probe_list=None, k=1, return_address=False,
                     _extents=None):
        """Scan the given cells [n_query, n_probe] for each query; (values, ids[, address]).
        (`_extents`: the cells' (start, size) when the coarse step already gathered them.)"""
        n_query = x.shape[1]
        if n_probe_list is None:
            n_probe_list = torch.full((n_query,), cells.shape[1], device=self.device, dtype=torch.long)
        if _extents is None:
            cell_start = self._cell_start[cells]
            cell_size = self._cell_size[cells]
        else:
            cell_start, cell_size = _extents
        # expected slots per query (host-side estimate, no sync): bounds the per-query split
        slots_hint = cells.shape[1] * self.capacity // max(self.n_cells, 1)
        if self.pq_use_residual:
            assert base_sims is not None, "base_sims is required when pq_use_residual is True"
            is_empty = self._is_empty if self._has_holes else None
            from ..kernels import PACKED_M
            if self.use_precomputed and self.use_packed_layout and self.n_subvectors in PACKED_M:
                # scan layout: part1[q] staged once per query (built in the workgroup while the
                # sub-vectors are short), the cell-dependent half folded into a per-slot constant
                if self._precomputed_part2 is None:
                    self.precompute_part2()
                slot_term, cell_bound = self._residual_slot_terms()
                fused = self.use_fused_lut and self.d_subvector <= self.fused_lut_max_subvector
                part1 = None if fused else self.precomputed_adc_residual_precomputed(x)[0]
                topk_val, topk_address, topk_ids = self._ivfpq_topk._scan.topk_residual_packed(
                    data=self._storage, packed=self.packed_storage(), part2=self._part2_by_cell,
                    slot_term=slot_term, cell_bound=cell_bound, cells=cells, base_sims=base_sims,
                    is_empty=is_empty, cell_start=cell_start, cell_size=cell_size,
                    n_probe_list=n_probe_list, n_candidates=k, part1=part1,
                    query=x if fused else None,
                    codebook=self.pq_codec.codebook if fused else None,
                    address2id=self._address2id, slots_hint=slots_hint)
            elif self.use_precomputed:
                part1, part2 = self.precomputed_adc_residual_precomputed(x)
                topk_val, topk_address, topk_ids = self._ivfpq_topk.topk_residual_precomputed(
                    data=self._storage, part1=part1, part2=part2, cells=cells, base_sims=base_sims,
                    cell_start=cell_start, cell_size=cell_size, is_empty=is_empty,
                    n_probe_list=n_probe_list, k=k, address2id=self._address2id)
            else:
                precomputed = self.precomputed_adc_residual(x, cells)
                topk_val, topk_address, topk_ids = self._ivfpq_topk.topk_residual(
                    data=self._storage, base_sims=base_sims, precomputed=precomputed,
                    cell_start=cell_start, cell_size=cell_size, is_empty=is_empty,
                    n_probe_list=n_probe_list, k=k, address2id=self._address2id)
            if return_address:
                return topk_val, topk_ids, topk_address
            return topk_val, topk_ids
        packed = None
        if self.use_packed_layout:
            from ..kernels import PACKED_M
            if self.n_subvectors in PACKED_M and (
                    self.n_subvectors >= self.packed_min_subvectors
                    or self.n_subvectors <= self.packed_max_short_subvectors):
                packed = self.packed_storage()
        # the fused path re-reads the codebook (m*ds KiB, L2-resident) per workgroup instead of a
        # 1-KiB-per-sub-quantizer LUT row from HBM: a win while the sub-vectors are short
        if self.use_fused_lut and self.d_subvector <= self.fused_lut_max_subvector:
            topk_val, topk_address, topk_ids = self._ivfpq_topk.topk_fused(
                data=self._storage, query=x, codebook=self.pq_codec.codebook, cell_start=cell_start,
                cell_size=cell_size, is_empty=self._is_empty if self._has_holes else None,
                n_probe_list=n_probe_list, k=k, distance=self.distance, packed=packed,
                address2id=self._address2id, slots_hint=slots_hint)
            if return_address:
                return topk_val, topk_ids, topk_address
            return topk_val, topk_ids
        precomputed = self.pq_codec.precompute_adc(x)
        topk_val, topk_address, topk_ids = self._ivfpq_topk.topk(
            data=self._storage, precomputed=precomputed, cell_start=cell_start,
            cell_size=cell_size, is_empty=self._is_empty if self._has_holes else None,
            n_probe_list=n_probe_list, k=k, packed=packed, address2id=self._address2id,
            slots_hint=slots_hint)
        if return_address:
            return topk_val, topk_ids, topk_address
        return topk_val, topk_ids

    def _probe_with_extents(self, x):
        """probe() plus the (start, size) of every probed cell, or None when not gathered"""
        if self.use_fused_probe and self.use_cublas and self.n_probe <= 1024:
            smart = self.use_smart_probing and self.n_probe > 1
            sims, cells, cs, sz, npl = self._coarse_probe(
                x, self.vq_codec.codebook, self._cell_start, self._cell_size, self.n_probe,
                self.smart_probing_temperature if smart else None, prepared=self._probe_prepared())
            return sims, cells, npl, (cs, sz)
        return (*self.probe(x), None)

    def _probe_prepared(self):
        """the coarse codebook's share of the fp16 selection pass, rebuilt when the codebook changes"""
        cb = self.vq_codec.codebook
        key = (cb.data_ptr(), tuple(cb.shape), util.tensor_version(cb))
        cached = getattr(self, "_probe_prep_cache", None)
        # (the entry HOLDS the codebook tensor and is matched by identity: a later codebook allocated at the freed
        # address with the same shape and version -- train, search, train, train, search -- must not hit it)
        if cached is None or cached[0] != key or cached[2] is not cb or cb.is_inference():
            self._probe_prep_cache = cached = (key, self._coarse_probe.prepare(cb), cb)
        return cached[1]

    def probe(self, x):
        """Coarse step: (topk_sims, cells [n_query, n_probe], n_probe_list [n_query])."""
        if self.use_fused_probe and self.use_cublas and self.n_probe <= 1024:
            return self._probe_with_extents(x)[:3]
        vq_codebook = self.vq_codec.codebook
        if self.use_cublas and self.n_probe <= 1024:
            # library GEMM, then the 2ab - a^2 - b^2 epilogue (reference rounding order) fused into
            # the row select: one pass over the [n_query, n_cells] matrix instead of four
            dots = x.transpose(0, 1).contiguous() @ vq_codebook
            topk_sims, cells = self._coarse_select(dots, (x * x).sum(dim=0),
                                                   (vq_codebook * vq_codebook).sum(dim=0), self.n_probe)
        elif self.use_cublas:
            sims = metric.negative_squared_l2_distance(x, vq_codebook).contiguous()
            topk_sims, cells = self._topk(sims, k=self.n_probe, dim=1)
        else:
            topk_sims, cells = self.vq_codec.kmeans.topk(x, k=self.n_probe)
        if self.use_smart_probing and self.n_probe > 1:
            n_probe_list = self._smart_probing(topk_sims, self.smart_probing_temperature)
        else:
            n_probe_list = torch.full((x.shape[1],), self.n_probe, device=self.device,
                                      dtype=torch.long)
        return topk_sims, cells, n_probe_list

    def graphed_search(self, n_query, k=1):
        """search() for a fixed batch shape captured in one HIP graph (low-latency serving)"""
        from .graphed import GraphedSearch
        return GraphedSearch(self, n_query, k)

    def search(self, x, k=1, return_address=False):
        """x [d_vector, n_query] f32 -> (values f32 [n_query, k] descending, ids int64 [n_query, k]);
        values are -squared-L2 (euclidean) or cosine similarity of the PQ reconstruction.
        `return_address` is accepted and ignored, as in the reference (:521)."""
        assert len(x.shape) == 2
        assert x.shape[0] == self.d_vector
        assert 0 < k <= 1024
        assert self.vq_codec.is_trained and self.pq_codec.is_trained, "index is not trained"
        assert 1 <= self.n_probe <= self.n_cells
        x = x.to(self.device)
        if self.distance == "cosine":
            x = util.normalize(x, dim=0)
        n_query = x.shape[1]
        vals, ids = [], []
        for q0 in range(0, max(n_query, 1), self.max_query_batch):
            xb = x[:, q0:q0 + self.max_query_batch].contiguous()
            topk_sims, cells, n_probe_list, extents = self._probe_with_extents(xb)
            v, i = self.search_cells(x=xb, cells=cells, base_sims=topk_sims,
                                     n_probe_list=n_probe_list, k=k, return_address=False,
                                     _extents=extents)
            vals.append(v)
            ids.append(i)
        if len(vals) == 1:
            return vals[0], ids[0]
        return torch.cat(vals, 0), torch.cat(ids, 0)
