"""3-NN feature interpolation (models/layers/upsampling.py:11-102)."""
from amcontrast3d_amd.ops import three_interpolate, three_interpolation, three_nn  # noqa: F401
