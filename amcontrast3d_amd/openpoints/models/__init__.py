from .build import MODELS, build_model_from_cfg
from .backbone import *  # noqa: F401,F403
from .segmentation import *  # noqa: F401,F403
from openpoints.AMContrast3D.APM import APM_pf_ConCate  # noqa: F401,E402  (registers the APM)
