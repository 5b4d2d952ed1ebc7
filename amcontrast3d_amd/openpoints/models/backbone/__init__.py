from .pointnext_AA import (FeaturePropogation, InvResMLP, LocalAggregation, PointNextDecoder_AMContrast3D,
                           PointNextEncoder_AMContrast3D, ResBlock, SetAbstraction)
