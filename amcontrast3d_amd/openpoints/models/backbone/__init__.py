from .pointnext_AA import (FeaturePropogation, InvResMLP, LocalAggregation, PointNextDecoder_AMContrast3D,
                           PointNextEncoder_AMContrast3D, ResBlock, SetAbstraction)
from .pointnext_MM import PointNextDecoder_M_AMContrast3D, PointNextEncoder_M_AMContrast3D
