"""pointops.knnquery on the gfx950 kernel (cpp/pointops/functions/pointops.py:32-56).

Only ``knnquery`` is reached from the AMContrast3D path;
the other pointops entry points of the reference have no caller in this fork.
"""
from amcontrast3d_amd.ops import KNNQuery, knnquery  # noqa: F401
