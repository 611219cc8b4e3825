"""ctypes binding of libtorchpq_amd.so (include/torchpq_amd.h).

The HIP library is the product: there is NO CPU fallback.  If the shared object is missing
or a call fails, this module raises -- loudly -- instead of routing around it.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# TPQ_AMD_LIB: load an experimental build of the SAME library instead (tools/build_variant.sh;
# kernel A/B measurements only -- there is still no fallback of any kind)
LIB_PATH = os.environ.get("TPQ_AMD_LIB") or os.path.join(_HERE, "libtorchpq_amd.so")

METRIC_NEG_SQ_L2 = 0
METRIC_INNER = 1
ERR_UNSUPPORTED = -4  # TPQ_ERR_UNSUPPORTED
ASSIGN_ROUTE_AUTO = 0     # TPQ_ASSIGN_ROUTE_AUTO
ASSIGN_ROUTE_CASCADE = 1  # TPQ_ASSIGN_ROUTE_CASCADE
PROBE_ROUTE_AUTO, PROBE_ROUTE_FP32, PROBE_ROUTE_FP16 = 0, 1, 2  # TPQ_PROBE_ROUTE_*

_vp, _i, _i64, _sz, _f = C.c_void_p, C.c_int, C.c_int64, C.c_size_t, C.c_float

# name -> (restype, argtypes); mirrors include/torchpq_amd.h one to one
SIGNATURES = {
    "tpq_version": (_i, []),
    "tpq_last_error": (C.c_char_p, []),
    "tpq_ivfpq_scan_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "tpq_ivfpq_scan_topk": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i,
                                 _i, _i, _vp, _sz, _vp]),
    "tpq_ivfpq_pack_codes": (_i, [_vp, _vp, _i64, _i, _i64, _i64, _vp]),
    "tpq_ivfpq_scan_topk_packed": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                        _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "tpq_ivfpq_search_fused": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                    _i64, _i, _i, _i, _i, _i, _vp, _sz, _vp]),
    "tpq_ivfpq_scan_tickets_bytes": (_sz, [_i]),
    "tpq_ivfpq_scan_route": (_i, [_i, _i, _i, _i, _i, _i, _i64, _i, _i, _i, _i]),
    "tpq_ivfpq_scan_topk_packed_tickets": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64,
                                                _i, _i, _i, _i, _i, _vp, _sz, _vp, _i64, _vp]),
    "tpq_ivfpq_search_fused_tickets": (_i, [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                            _i64, _i, _i, _i, _i, _i, _vp, _sz, _vp, _i64, _vp]),
    "tpq_ivfpq_scan_topk_residual": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _vp, _i64, _i, _i, _i, _i, _vp]),
    "tpq_residual_part1": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "tpq_ivfpq_residual_slot_terms": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _vp]),
    "tpq_ivfpq_scan_topk_residual_packed": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp,
                                                 _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i,
                                                 _i, _i, _i, _vp, _sz, _vp]),
    "tpq_adc_lut": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "tpq_topk_select": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp]),
    "tpq_coarse_select": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "tpq_smart_probing": (_i, [_vp, _vp, _i, _i, _f, _vp]),
    "tpq_ivfpq_coarse_probe_workspace_bytes": (_sz, [_i, _i]),
    "tpq_ivfpq_coarse_probe": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f,
                                    _vp, _sz, _vp]),
    "tpq_ivfpq_coarse_probe_route_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "tpq_ivfpq_coarse_probe_route": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _i,
                                          _vp, _vp, _sz, _vp]),
    "tpq_ivfpq_coarse_probe_prepared_bytes": (_sz, [_i, _i]),
    "tpq_ivfpq_coarse_probe_prepare": (_i, [_vp, _i, _i, _vp, _sz, _vp]),
    "tpq_get_id_by_address": (_i, [_vp, _i64, _vp, _vp, _i64, _vp]),
    "tpq_get_address_by_id": (_i, [_vp, _i64, _vp, _vp, _i64, _vp]),
    "tpq_max_sim": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tpq_max_sim_split_supported": (_i, [_i, _i64, _i]),
    "tpq_max_sim_split": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "tpq_max_sim_select_supported": (_i, [_i, _i, _i64, _i]),
    "tpq_max_sim_select_workspace_bytes": (_sz, [_i, _i, _i64, _i]),
    "tpq_max_sim_select": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i64, _i, _i, _vp, _sz, _vp]),
    "tpq_lloyd_supported": (_i, [_i, _i, _i64, _i]),
    "tpq_lloyd_prepared_bytes": (_sz, [_i, _i, _i64]),
    "tpq_lloyd_prepare": (_i, [_vp, _vp, _vp, _sz, _i, _i, _i64, _i, _vp]),
    "tpq_lloyd_step_workspace_bytes": (_sz, [_i, _i, _i64, _i]),
    "tpq_lloyd_step_count_offset": (_sz, [_i, _i, _i64, _i, _i]),
    "tpq_lloyd_step": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _i, _vp, _sz, _vp]),
    "tpq_coarse_assign_supported": (_i, [_i, _i64, _i]),
    "tpq_coarse_assign_workspace_bytes": (_sz, [_i, _i64, _i]),
    "tpq_coarse_assign_count_offset": (_sz, [_i, _i64, _i]),
    "tpq_coarse_assign": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _vp, _sz, _vp]),
    "tpq_coarse_assign_route_workspace_bytes": (_sz, [_i, _i64, _i, _i]),
    "tpq_coarse_assign_route": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i, _i, _i, _vp, _sz, _vp]),
    "tpq_compute_centroids_workspace_bytes": (_sz, [_i, _i, _i]),
    "tpq_compute_centroids": (_i, [_vp, _vp, _vp, _i, _i, _i64, _i, _vp, _sz, _vp]),
    "tpq_get_ioa_workspace_bytes": (_sz, [_i64]),
    "tpq_get_ioa": (_i, [_vp, _vp, _i64, _i64, _vp, _sz, _vp]),
    "tpq_get_write_address": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "tpq_get_cell_by_address": (_i, [_vp, _vp, _vp, _vp, _i64, _i64, _vp]),
    "tpq_grow_cells": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i, _i, _vp]),
    "tpq_pq_decode": (_i, [_vp, _vp, _vp, _i, _i, _i64, _vp]),
    "tpq_scatter_codes": (_i, [_vp, _vp, _vp, _vp, _i, _i64, _i64, _vp]),
    "tpq_ubench_stream_read": (_i, [_vp, _sz, _vp, _i, _vp]),
    "tpq_ubench_stream_read_ex": (_i, [_vp, _sz, _vp, _i, _i, _i, _sz, _i, _vp]),
    "tpq_ubench_rows_read": (_i, [_vp, _i, _i, _i64, _i, _vp, _vp]),
}

_lib = None


class TorchPQAmdError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if the library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TorchPQAmdError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or torchpq_amd/csrc/build.sh). torchpq_amd has no CPU fallback.")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def last_error() -> str:
    msg = load().tpq_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc: int, what: str):
    if rc != 0:
        raise TorchPQAmdError(f"{what} failed (code {rc}): {last_error()}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """The CURRENT torch stream at call time (the reference captures the stream once at
    wrapper construction, kernels/CustomKernel.py:16 -- a quirk not reproduced)."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def require_gpu(*tensors):
    for t in tensors:
        if t is None:
            continue
        if t.device.type != "cuda":
            raise TorchPQAmdError(
                "torchpq_amd runs on an AMD GPU (torch device 'cuda'); got a tensor on "
                f"'{t.device}'. There is no CPU fallback.")
        if not t.is_contiguous():
            raise TorchPQAmdError("torchpq_amd kernels need contiguous tensors")
