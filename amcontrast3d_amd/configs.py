"""Model / loss configurations of the AMContrast3D-AA path as plain dicts.

The reference ships only the XL model (cfgs/s3dis/AMContrast3D-AA.yaml:32-64, blocks
[1,4,7,4,4], width 64, sa_layers 1); PointNeXt-S / -B / -L named by BASELINE.json are
the standard PointNeXt variants under the same schema:
S = blocks [1,1,1,1,1], width 32, sa_layers 2, sa_use_res True; L = [1,3,5,3,3],
width 32.  ``ambiguity_args`` are the values of cfgs/{s3dis,scannet}/AMContrast3D-AA.yaml:6-30.
Feed the dicts to ``openpoints.utils.EasyConfig().update(...)``.
"""
import copy

_VARIANTS = {
    #        blocks            width sa_layers sa_use_res
    'S': ([1, 1, 1, 1, 1], 32, 2, True),
    'B': ([1, 2, 3, 2, 2], 32, 1, False),
    'L': ([1, 3, 5, 3, 3], 32, 1, False),
    'XL': ([1, 4, 7, 4, 4], 64, 1, False),
}


def model_cfg(variant='S', num_classes=13, in_channels=4, dropout=0.5, radius=0.1, nsample=32, width=None,
              blocks=None, global_feat=None):
    b, w, sa_layers, sa_use_res = _VARIANTS[variant]
    cls_args = {'NAME': 'SegHead', 'num_classes': num_classes, 'in_channels': None, 'norm_args': {'norm': 'bn'},
                'dropout': dropout}
    if global_feat is not None:
        cls_args['global_feat'] = global_feat
    return copy.deepcopy({
        'NAME': 'BaseSeg_AMContrast3D',
        'encoder_args': {
            'NAME': 'PointNextEncoder_AMContrast3D',
            'blocks': list(blocks if blocks is not None else b),
            'strides': [1, 4, 4, 4, 4],
            'sa_layers': sa_layers,
            'sa_use_res': sa_use_res,
            'width': width if width is not None else w,
            'in_channels': in_channels,
            'expansion': 4,
            'radius': radius,
            'nsample': nsample,
            'aggr_args': {'feature_type': 'dp_fj', 'reduction': 'max'},
            'group_args': {'NAME': 'ballquery', 'normalize_dp': True},
            'conv_args': {'order': 'conv-norm-act'},
            'act_args': {'act': 'relu'},
            'norm_args': {'norm': 'bn'},
        },
        'decoder_args': {'NAME': 'PointNextDecoder_AMContrast3D'},
        'cls_args': cls_args,
    })


def model_cfg_mm(variant='XL', num_classes=13, in_channels=4, dropout=0.5, width=None, blocks=None, ignore_index=None,
                 dataset='s3dis', **apm):
    """AMContrast3D++ (cfgs/{s3dis,scannet}/AMContrast3D-MM.yaml:33-88): the AA model plus the ambiguity
    prediction module and masked refinement.  `apm`: overrides of the APM_args block."""
    cfg = model_cfg(variant, num_classes=num_classes, in_channels=in_channels, dropout=dropout, width=width, blocks=blocks)
    w = cfg['encoder_args']['width']
    cfg['NAME'] = 'BaseSeg_M_AMContrast3D'
    cfg['encoder_args']['NAME'] = 'PointNextEncoder_M_AMContrast3D'
    cfg['decoder_args']['NAME'] = 'PointNextDecoder_M_AMContrast3D'
    cfg['cls_args']['ignore_index'] = ignore_index
    cfg['AEF_args'] = ambiguity_args_mm(dataset)
    apm_args = {'NAME': 'APM_pf_ConCate', 'feature_dim': [w, 2 * w, 4 * w, 8 * w], 'linear_mapping': False,
                'cross_attention': False, 'feat_concate': False, 'channel': [32, 16, 8, 4, 2], 'dropout': [0, 0, 0, 0, 0],
                'nsample_k': 12, 'threshold': 0.9, 'threshold_max': 1.0, 'gamma': 1, 'fusion': 'MIN', 'att_dim': 3}
    apm_args.update(apm)
    cfg['APM_args'] = apm_args
    return copy.deepcopy(cfg)


def ambiguity_args_mm(dataset='s3dis'):
    """ambiguity_args of the MM configs: the AA values plus the regression weight and the refinement's source"""
    args = ambiguity_args(dataset)
    args.update({'w3': 0.01, 'source': 'APM', 'source_mode': 'Train'})
    return args


def criterion_cfg_mm():
    return {'NAME': 'CrossEntropyAcePre'}


def ambiguity_args(dataset='s3dis'):
    args = {
        'action': False, 'vis': False, 'nsample': 24, 'ccbeta': 0.04, 'cctype': 'Method2',
        'temperature': 0.3, 'supervisedCL': 'Method1', 'db': '-m', 'margin': 'adaptive',
        'mu': -1, 'nu': 0.5, 'miou_B_I': False, 'w1': 0.1, 'w2': 0.9, 'stages': 'up', 'stages_num': 4,
    }
    if dataset == 'scannet':  # cfgs/scannet/AMContrast3D-AA.yaml
        args.update({'temperature': 0.5, 'nu': 0.6})
    return args


def criterion_cfg():
    # cfgs/s3dis/default.yaml:59-61 (label_smoothing is accepted and ignored by CrossEntropyAce)
    return {'NAME': 'CrossEntropyAce', 'label_smoothing': 0.2}
