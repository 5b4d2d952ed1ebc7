"""The model registry of the drop-in package.

`MODELS` maps a class name to the class (the `@MODELS.register_module()` decorators of backbone/ and segmentation/
fill it on import); `build_model_from_cfg(cfg, **default_args)` looks up `cfg.NAME` and calls the class with the
remaining keys of `cfg` as keyword arguments -- the entry point examples/segmentation/main_AA.py:142 uses
(reference: openpoints/models/build.py:1-13, openpoints/utils/registry.py:8-294).
"""
from openpoints.utils.registry import Registry

MODELS = Registry('models')


def build_model_from_cfg(cfg, **kwargs):
    """Build the model named by ``cfg.NAME`` (same parameter names as the reference: callers may pass ``cfg=``)."""
    return MODELS.build(cfg, **kwargs)
