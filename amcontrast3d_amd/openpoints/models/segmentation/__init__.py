from .base_seg import BaseSeg_AMContrast3D, BaseSeg_M_AMContrast3D, SegHead
