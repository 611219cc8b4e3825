"""Optional import alias: make ``import torchpq`` resolve to this package.

    import torchpq_amd.compat as compat
    compat.install_as_torchpq()
    from torchpq.index import IVFPQIndex            # the MI355X implementation
    from torchpq.kernels import IVFPQTopkCuda       # -> IVFPQTopkHip (same call signature)

Only the names on the IVFPQ train / add / search path exist; everything else
of the reference (legacy, transform, experimental, the distributed containers) raises
AttributeError on access, loudly, rather than half-working.  Nothing is installed implicitly:
a process that also has the real TorchPQ importable must choose.
"""
import importlib
import sys
import types

# reference wrapper name (torchpq/kernels/__init__.py) -> class in torchpq_amd.kernels
KERNEL_ALIASES = {
    "IVFPQTopkCuda": "IVFPQTopkHip",
    "IVFPQTop1Cuda": "IVFPQTop1Hip",
    "MaxSimCuda": "MaxSimHip",
    "ComputeCentroidsCuda": "ComputeCentroidsHip",
    "TopkSelectCuda": "TopkSelectHip",
    "Top32SelectCuda": "Top32SelectHip",
    "Top1SelectCuda": "Top1SelectHip",
    "GetIOACuda": "GetIOAHip",
    "GetWriteAddressV2Cuda": "GetWriteAddressHip",
    "GetDivByAddressV2Cuda": "GetCellByAddressHip",
    "PQDecodeCuda": "PQDecodeHip",
}
SUBMODULES = ("index", "codec", "clustering", "container", "fn", "kernels", "metric", "util",
              "CustomModule")


def install_as_torchpq(force=False):
    """Register ``torchpq`` (and its sub-modules on the hot path) in sys.modules as aliases of
    torchpq_amd.  Refuses to shadow an already imported ``torchpq`` unless force=True."""
    existing = sys.modules.get("torchpq")
    if existing is not None and not getattr(existing, "__torchpq_amd_alias__", False) and not force:
        raise RuntimeError("a different `torchpq` is already imported; pass force=True to shadow it")
    import torchpq_amd
    top = types.ModuleType("torchpq", "alias of torchpq_amd (MI355X-native IVFPQ path)")
    top.__torchpq_amd_alias__ = True
    top.__path__ = []  # a package, but with no files of its own
    top.__version__ = getattr(torchpq_amd, "__version__", "0")
    sys.modules["torchpq"] = top
    for name in SUBMODULES:
        mod = importlib.import_module("torchpq_amd." + name)
        if name == "kernels":
            mod = _kernels_alias(mod)
        sys.modules["torchpq." + name] = mod
        setattr(top, name, mod)
    top.CustomModule = sys.modules["torchpq.CustomModule"].CustomModule
    top.topk = sys.modules["torchpq.fn"].Topk()
    return top


def _kernels_alias(kernels):
    alias = types.ModuleType("torchpq.kernels", "reference wrapper names -> HIP wrappers")
    for name in getattr(kernels, "__all__", []):
        setattr(alias, name, getattr(kernels, name))
    for ref_name, hip_name in KERNEL_ALIASES.items():
        setattr(alias, ref_name, getattr(kernels, hip_name))
    return alias


def uninstall():
    for key in [k for k in sys.modules if k == "torchpq" or k.startswith("torchpq.")]:
        mod = sys.modules[key]
        if key == "torchpq" and not getattr(mod, "__torchpq_amd_alias__", False):
            return
        del sys.modules[key]
