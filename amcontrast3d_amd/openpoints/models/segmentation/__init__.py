from .base_seg import BaseSeg_AMContrast3D, SegHead
