"""ORACLE -- test infrastructure only; runs ONLY in the build container.

Imports the reference's own Python layer (read-only, from /root/reference) on a
machine without CUDA so that oracle/gen_golden.py can record golden vectors:

  * third-party modules the image lacks and the reference imports at module
    scope (easydict, multimethod, termcolor, shortuuid, h5py, ...) get minimal
    in-process stand-ins;
  * the two CUDA extension modules are replaced by the C restatement
    (oracle/pointops_ref.py) under their real names
    ``pointnet2_batch_cuda`` / ``pointops_cuda``;
  * ``torch.cuda.IntTensor/FloatTensor``, ``Tensor.cuda()``, ``Module.cuda()`` and
    ``device='cuda'`` are mapped to the CPU.

Nothing from /root/reference is copied; nothing here is shipped or imported by
the product.  /root/reference does not exist on the GPU box: only the .npz
fixtures this produces travel.
"""
import sys
import types

import torch

REFERENCE_ROOT = "/root/reference"


def _stub_third_party():
    if "easydict" not in sys.modules:
        m = types.ModuleType("easydict")

        class EasyDict(dict):
            def __init__(self, d=None, **kw):
                super().__init__()
                for k, v in dict(d or {}, **kw).items():
                    self[k] = v

            def __setitem__(self, k, v):
                if isinstance(v, dict) and not isinstance(v, EasyDict):
                    v = EasyDict(v)
                super().__setitem__(k, v)

            def __getattr__(self, k):
                try:
                    return self[k]
                except KeyError:
                    raise AttributeError(k)

            __setattr__ = __setitem__

        m.EasyDict = EasyDict
        sys.modules["easydict"] = m

    if "multimethod" not in sys.modules:
        m = types.ModuleType("multimethod")
        import typing

        def _matches(value, ann):
            origin = typing.get_origin(ann)
            if ann is typing.Dict or origin is dict:
                return isinstance(value, dict)
            if origin is typing.Union:
                return any(_matches(value, a) for a in typing.get_args(ann))
            if ann is typing.List or origin is list:
                return isinstance(value, list)
            if ann is typing.Tuple or origin is tuple:
                return isinstance(value, tuple)
            return isinstance(value, ann) if isinstance(ann, type) else True

        _registry = {}

        def multimethod(fn):
            key = (fn.__module__, fn.__qualname__)
            impls = _registry.setdefault(key, [])
            impls.append(fn)

            def dispatch(self, arg, *a, **kw):
                for f in impls:
                    ann = [v for k, v in f.__annotations__.items() if k != "return"]
                    if not ann or _matches(arg, ann[0]):
                        return f(self, arg, *a, **kw)
                raise TypeError("no overload for %r" % type(arg))

            return dispatch

        m.multimethod = multimethod
        sys.modules["multimethod"] = m

    for name, attrs in (("termcolor", {"colored": lambda s, *a, **k: s}),
                        ("shortuuid", {"uuid": lambda: "oracle"}),
                        ("h5py", {}), ("wandb", {}), ("torch_scatter", {"scatter": None})):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                mod = types.ModuleType(name)
                for k, v in attrs.items():
                    setattr(mod, k, v)
                sys.modules[name] = mod


def _cpu_cuda_shims():
    def _int_tensor(*shape, **kw):
        kw.pop("device", None)
        if len(shape) == 1 and isinstance(shape[0], (list, tuple)):
            return torch.tensor(shape[0], dtype=torch.int32)
        return torch.zeros(*shape, dtype=torch.int32)

    def _float_tensor(*shape, **kw):
        kw.pop("device", None)
        if len(shape) == 1 and isinstance(shape[0], (list, tuple)):
            return torch.tensor(shape[0], dtype=torch.float32)
        return torch.zeros(*shape, dtype=torch.float32)

    torch.cuda.IntTensor = _int_tensor
    torch.cuda.FloatTensor = _float_tensor
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self

    def _strip_device(fn):
        def wrapped(*a, **kw):
            if "device" in kw and str(kw["device"]).startswith("cuda"):
                kw["device"] = "cpu"
            return fn(*a, **kw)
        return wrapped

    for name in ("zeros", "ones", "full", "empty", "tensor", "arange", "rand", "randn"):
        setattr(torch, name, _strip_device(getattr(torch, name)))


def load_reference():
    """Returns the imported reference ``openpoints`` package (models + loss registries populated)."""
    import os
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.dirname(here))
    from oracle import pointops_ref
    pointops_ref.build()
    _stub_third_party()
    _cpu_cuda_shims()
    m1, m2 = pointops_ref.as_modules()
    sys.modules["pointnet2_batch_cuda"] = m1
    sys.modules["pointops_cuda"] = m2
    if "openpoints" in sys.modules:
        raise RuntimeError("an 'openpoints' package is already imported; run the generator in a fresh process")
    sys.path.insert(0, REFERENCE_ROOT)
    import openpoints  # noqa: F401  (the reference's)
    import openpoints.models  # noqa: F401
    import openpoints.loss  # noqa: F401
    assert openpoints.__file__.startswith(REFERENCE_ROOT)
    return openpoints
