"""Import the reference (DeMoriarty/TorchPQ) Python package in THIS container.

TEST INFRASTRUCTURE ONLY; build-container only (``/root/reference`` does not
exist on the GPU box).  The reference hard-requires CuPy (torchpq/__init__.py:2-5)
and a CUDA stream at kernel-wrapper construction (kernels/CustomKernel.py:16),
neither of which exists here, so a stub ``cupy`` whose RawKernel raises when
launched is injected before the import.  Only the reference's own CPU/PyTorch
code paths can then run -- which is exactly what is used to pin the oracle.
"""
from __future__ import annotations

import os
import sys
import types

REFERENCE_ROOT = os.environ.get("TORCHPQ_REFERENCE", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "torchpq"))


def import_reference():
    """Returns the imported reference ``torchpq`` module (CPU paths only)."""
    if "torchpq" in sys.modules and getattr(sys.modules["torchpq"], "__graft_stub__", False):
        return sys.modules["torchpq"]
    assert available(), f"reference not found at {REFERENCE_ROOT}"
    sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only tree

    import torch

    class _RawKernel:
        def __init__(self, *a, **k):
            self.max_dynamic_shared_size_bytes = 0
            self.attributes = {}

        def __call__(self, *a, **k):
            raise RuntimeError("stub cupy: device kernels cannot run in this container")

    class _Device:
        def __init__(self, *a, **k):
            self.id = 0

        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def use(self):
            pass

    cp = types.ModuleType("cupy")
    cp.RawKernel = _RawKernel
    cp.memoize = lambda *a, **k: (lambda f: f)
    cp.cuda = types.ModuleType("cupy.cuda")
    cp.cuda.Device = _Device
    cp.cuda.set_allocator = lambda *a, **k: None
    cp.cuda.compile_with_cache = lambda *a, **k: None
    cp.cuda.memory = types.ModuleType("cupy.cuda.memory")
    cp.cuda.memory.UnownedMemory = object
    cp.cuda.memory.MemoryPointer = object
    cp.cuda.MemoryPointer = object
    cp.cuda.UnownedMemory = object
    cp.cuda.Stream = object
    cp.cuda.ExternalStream = lambda *a, **k: None
    sys.modules.setdefault("cupy", cp)
    sys.modules.setdefault("cupy.cuda", cp.cuda)
    sys.modules.setdefault("cupy.cuda.memory", cp.cuda.memory)

    class _Stream:
        cuda_stream = 0

    if not torch.cuda.is_available():
        torch.cuda.current_stream = lambda *a, **k: _Stream()  # CustomKernel.py:16
        torch.cuda.current_device = lambda *a, **k: 0

    sys.path.insert(0, REFERENCE_ROOT)
    try:
        import torchpq  # noqa: F401
    finally:
        sys.path.remove(REFERENCE_ROOT)
    torchpq.__graft_stub__ = True
    return torchpq
