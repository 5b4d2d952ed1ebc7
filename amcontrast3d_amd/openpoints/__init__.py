"""MI355X-native drop-in for the model / loss side of the reference's ``openpoints`` package.

Import it as the top-level package ``openpoints`` (``amcontrast3d_amd.activate()`` puts
its parent directory on sys.path).  If ``AMC3D_REFERENCE_ROOT`` points at a checkout of
the reference, the sub-packages this build does not provide (dataset, transforms,
optim, scheduler and the trainer-side utils) resolve from there, so
``examples/segmentation/main_AA.py`` runs unchanged on top of these models, losses
and kernels.
"""
import os as _os

_ref = _os.environ.get('AMC3D_REFERENCE_ROOT')
if _ref:
    _their = _os.path.join(_ref, 'openpoints')
    if _os.path.isdir(_their) and _their not in __path__:
        __path__.append(_their)
        from . import utils as _utils
        _their_utils = _os.path.join(_their, 'utils')
        if _os.path.isdir(_their_utils) and _their_utils not in _utils.__path__:
            _utils.__path__.append(_their_utils)
            _utils._overlay_optional()
