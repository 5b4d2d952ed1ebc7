from .blocks import (CHANNEL_MAP, Conv1d, Conv2d, create_act, create_convblock1d, create_convblock2d,
                     create_norm, feature_propagation_first_block, fused_first_block, fused_first_conv,
                     fused_local_aggregation, run_convblocks)
from .group import (ball_query, create_grouper, gather_operation, get_aggregation_feautres,
                    grouping_operation, torch_grouping_operation, QueryAndGroup, GroupAll)
from .subsample import fps, furthest_point_sample, random_sample
from .upsampling import three_interpolate, three_interpolation, three_nn
