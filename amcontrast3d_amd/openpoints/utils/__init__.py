"""openpoints.utils -- the part the model/loss path needs (registry, config).

When the reference tree is overlaid (see amcontrast3d_amd.activate), the
remaining helper modules the trainer imports (logger, ckpt_util, wandb, ...)
resolve from the reference's own openpoints/utils directory.
"""
from . import registry
from .config import EasyConfig
from .metrics import AverageMeter, ConfusionMatrix, get_mious
from .registry import Registry, build_from_cfg
from .ckpt_util import (cal_model_parm_nums, get_missing_parameters_message, get_unexpected_parameters_message,  # noqa: F401
                        load_checkpoint, resume_checkpoint, resume_model, resume_optimizer, save_checkpoint)


def _overlay_optional():
    # names main_AA.py imports from openpoints.utils (examples/segmentation/main_AA.py:14-16);
    # present only when the reference's utils directory is on this package's __path__
    import importlib
    table = {
        'random': ['set_random_seed'],
        'logger': ['setup_logger_dist', 'generate_exp_directory', 'resume_exp_directory'],
        'wandb': ['Wandb'],
        'dist_utils': ['reduce_tensor', 'gather_tensor', 'find_free_port'],
    }
    for mod, names in table.items():
        try:
            m = importlib.import_module(f'{__name__}.{mod}')
        except Exception:
            continue
        for n in names:
            if hasattr(m, n):
                globals()[n] = getattr(m, n)


if len(__path__) > 1:
    _overlay_optional()
