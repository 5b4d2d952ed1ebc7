"""MODELS registry (models/build.py:1-13)."""
from openpoints.utils import registry

MODELS = registry.Registry('models')


def build_model_from_cfg(cfg, **kwargs):
    """Build the model named by ``cfg.NAME``; remaining keys are ctor kwargs."""
    return MODELS.build(cfg, **kwargs)
