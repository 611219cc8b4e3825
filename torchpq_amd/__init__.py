"""torchpq_amd -- an MI355X-native IVFPQ search path behind the TorchPQ IVFPQIndex API.

Only the IVFPQ train / add / search path of DeMoriarty/TorchPQ is provided;
all device work runs in hand-written HIP kernels for gfx950 (libtorchpq_amd.so, C ABI in
include/torchpq_amd.h).  There is no CPU fallback.
"""
from . import clustering, codec, container, fn, index, kernels, metric, util  # noqa: F401
from .CustomModule import CustomModule  # noqa: F401
from ._lib import TorchPQAmdError, load as load_library  # noqa: F401

from ._version import __version__  # noqa: F401,E402
