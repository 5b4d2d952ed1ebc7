from .build import MODELS, build_model_from_cfg
from .backbone import *  # noqa: F401,F403
from .segmentation import *  # noqa: F401,F403
# registers the APM variants (the reference does it from models/backbone/__init__.py:7-10)
from openpoints.AMContrast3D.APM import (APM_p, APM_p_Graph, APM_p_Group, APM_pf_ConCate, APM_pf_CrossAtt,  # noqa: F401,E402
                                         APM_pp_SelfAtt)
