from .concatenation import APM_pf_ConCate  # noqa: F401
from .attention import APM_pf_CrossAtt, APM_pp_SelfAtt  # noqa: F401
from .separation import APM_p, APM_p_Group, APM_p_Graph  # noqa: F401
