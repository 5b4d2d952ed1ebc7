"""MI355X-native PointNeXt SetAbstraction / FeaturePropagation stack + AMContrast3D loss.

Layout
    csrc/         hand-written gfx950 HIP kernels + the C-ABI (include/amc3d.h) -> libamc3d_hip.so
    _lib.py       ctypes loader (no fallback: missing library == ImportError)
    ops.py        torch.autograd front-ends mirroring the reference's wrappers
    compat.py     module objects shaped like the reference's pybind11 extensions
    openpoints/   drop-in for the reference's openpoints.{models,loss,AMContrast3D,utils.registry}
    synthetic.py  seeded S3DIS-shaped scenes for tests / bench
    dist.py       one-process-per-GPU data-parallel helpers (RCCL)
"""
import os
import sys

__version__ = "0.1.0"

_HERE = os.path.dirname(os.path.abspath(__file__))


def activate():
    """Make ``import openpoints`` resolve to this build's drop-in package."""
    if _HERE not in sys.path:
        sys.path.insert(0, _HERE)
    root = os.path.dirname(_HERE)
    if root not in sys.path:
        sys.path.insert(0, root)
    mod = sys.modules.get('openpoints')
    if mod is not None and not getattr(mod, '__file__', '').startswith(_HERE):
        raise ImportError("another 'openpoints' package is already imported: %s" % getattr(mod, '__file__', '?'))
