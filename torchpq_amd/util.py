"""Helpers mirroring torchpq/util.py (dtype helpers :8-36,82-84, normalize :38-43, the
shared-memory table :64-80 re-read for gfx950, tick :88-99)."""
from time import time

import torch

LDS_BYTES_PER_CU = 160 * 1024  # gfx950

_DTYPES = {
    "double": torch.float64, "float64": torch.float64, "half": torch.float16,
    "float16": torch.float16, "float": torch.float32, "float32": torch.float32,
    "bfloat16": torch.bfloat16, "long": torch.int64, "int64": torch.int64, "int": torch.int32,
    "int32": torch.int32, "int16": torch.int16, "int8": torch.int8, "uint8": torch.uint8,
    "bool": torch.bool,
}


def str2dtype(dtype_str):
    try:
        return _DTYPES[dtype_str]
    except KeyError:
        raise TypeError(f"Unrecognized dtype string: {dtype_str}")


def check_dtype(tensor, *dtype):
    wanted = [str2dtype(d) if isinstance(d, str) else d for d in dtype]
    return tensor.dtype in wanted


def check_device(tensor, *device):
    wanted = [torch.device(d) if isinstance(d, str) else d for d in device]
    return tensor.device in wanted


def tensor_version(t):
    """In-place-write counter of `t`, or None where PyTorch keeps none: tensors created under
    torch.inference_mode() raise on `._version`.  Callers that cache on (identity, data_ptr, version) then
    fall back to identity + data_ptr.  Inference tensors CAN be written in place inside inference mode, so the
    library does not rely on the counter for its own mutations: load_state_dict re-registers every buffer (a new
    tensor object), train() installs new tensors, add() / remove() / expand() bump CellContainer._codes_version
    (which GraphedSearch snapshots).  What goes unseen is a FOREIGN in-place write to a buffer under inference
    mode (e.g. `index._is_trained.fill_(...)`): do not do that to an index whose search is captured in a graph."""
    if t is None:
        return None
    return None if t.is_inference() else t._version


def normalize(x, dim=0):
    return x / (x.norm(dim=dim, keepdim=True) + 1e-9)


def get_maximum_shared_memory_bytes(device_id=0):
    """LDS a single workgroup may use.  The reference keys a table on the CUDA compute
    capability (util.py:64-80); every MI355X CU has 160 KiB."""
    return LDS_BYTES_PER_CU


def max_subvectors():
    """Largest n_subvectors whose fp32 LUT (1 KiB per sub-quantizer) plus the scan's
    bookkeeping fits one workgroup's LDS (reference gate: IVFPQIndex.py:28-29)."""
    return (LDS_BYTES_PER_CU - 8 * 1024) // 1024


_tm = -1.0


def tick(text="", init=False):
    global _tm
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if _tm < 0 or init:
        print(text, "initialized...")
    else:
        print(text, time() - _tm)
    _tm = time()
