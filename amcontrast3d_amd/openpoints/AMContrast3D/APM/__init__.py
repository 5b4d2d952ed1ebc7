from .concatenation import APM_pf_ConCate  # noqa: F401
