from .build import MODELS, build_model_from_cfg
from .backbone import *  # noqa: F401,F403
from .segmentation import *  # noqa: F401,F403
