from .build import MODELS, build_model_from_cfg
from .backbone import *  # noqa: F401,F403
from .segmentation import *  # noqa: F401,F403
# registers the APM variants (the reference does it from models/backbone/__init__.py:7-10).  A plain module import: when a
# caller's first import is an APM module itself, that package is only partly initialised at this point and finishes -- and
# registers its classes -- as soon as control returns to it; naming the classes here would fail on that order.
import openpoints.AMContrast3D.APM  # noqa: F401,E402
