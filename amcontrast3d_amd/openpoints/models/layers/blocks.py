"""1x1-conv / norm / activation block factories.

Same call signatures and module nesting as the reference
(models/layers/conv.py:8-102, norm.py:57-97, activation.py:5-51) so that
state-dict keys are identical: a block is ``nn.Sequential(conv[, norm][, act])``
-> ``<block>.0.weight`` (conv), ``<block>.1.{weight,bias,running_mean,...}`` (norm);
the conv has no bias when a norm follows it.
"""
import copy

import os

import torch
import torch.nn as nn


class Conv2d(nn.Conv2d):
    """nn.Conv2d that defaults to a 1x1 kernel when only (cin, cout) are given."""

    def __init__(self, *args, **kwargs):
        if len(args) == 2 and 'kernel_size' not in kwargs:
            args = (*args, (1, 1))
        super().__init__(*args, **kwargs)


class Conv1d(nn.Conv1d):
    """nn.Conv1d that defaults to kernel size 1 when only (cin, cout) are given."""

    def __init__(self, *args, **kwargs):
        if len(args) == 2 and 'kernel_size' not in kwargs:
            args = (*args, 1)
        super().__init__(*args, **kwargs)


_NORMS = {
    'bn1d': nn.BatchNorm1d, 'bn2d': nn.BatchNorm2d, 'bn': nn.BatchNorm2d,
    'in1d': nn.InstanceNorm1d, 'in2d': nn.InstanceNorm2d,
    'gn': nn.GroupNorm, 'syncbn': nn.SyncBatchNorm, 'ln': nn.LayerNorm,
}

_ACTS = {
    'silu': nn.SiLU, 'swish': nn.SiLU, 'mish': nn.Mish, 'relu': nn.ReLU, 'relu6': nn.ReLU6,
    'leaky_relu': nn.LeakyReLU, 'leakyrelu': nn.LeakyReLU, 'elu': nn.ELU, 'prelu': nn.PReLU,
    'celu': nn.CELU, 'selu': nn.SELU, 'gelu': nn.GELU, 'sigmoid': nn.Sigmoid, 'tanh': nn.Tanh,
    'hard_sigmoid': nn.Hardsigmoid, 'hard_swish': nn.Hardswish,
}


def create_norm(norm_args, channels, dimension=None):
    """norm.py:74-97: 'bn' + dimension '1d'/'2d' -> BatchNorm1d/2d; extra keys are ctor kwargs."""
    if norm_args is None:
        return None
    if isinstance(norm_args, dict):
        kwargs = copy.deepcopy(dict(norm_args))
        norm = kwargs.pop('norm', None)
    else:
        norm, kwargs = norm_args, {}
    if norm is None:
        return None
    if isinstance(norm, str):
        norm = norm.lower()
        if dimension is not None:
            dimension = str(dimension).lower()
            if dimension not in norm:
                norm += dimension
        assert norm in _NORMS, f"input {norm} is not supported"
        norm = _NORMS[norm]
    return norm(channels, **kwargs)


def create_act(act_args):
    """activation.py:25-51: in-place by default (except gelu/sigmoid)."""
    if act_args is None:
        return None
    act_args = {'act': act_args} if isinstance(act_args, str) else copy.deepcopy(dict(act_args))
    act = act_args.pop('act', None)
    if act is None:
        return None
    if isinstance(act, str):
        act = act.lower()
        assert act in _ACTS, f"input {act} is not supported"
        layer = _ACTS[act]
    inplace = act_args.pop('inplace', True)
    if act in ('gelu', 'sigmoid'):
        return layer(**act_args)
    return layer(inplace=inplace, **act_args)


def _convblock(conv_cls, dim, args, norm_args, act_args, order, kwargs):
    cin, cout = args[0], args[1]
    bias = kwargs.pop('bias', True)
    if order not in ('conv-norm-act', 'norm-act-conv', 'conv-act-norm'):
        raise NotImplementedError(f"{order} is not supported")
    norm = create_norm(norm_args, cin if order == 'norm-act-conv' else cout, dimension=dim)
    if norm is not None:
        bias = False
    conv = conv_cls(*args, bias=bias, **kwargs)
    act = create_act(act_args)
    parts = {'conv': conv, 'norm': norm, 'act': act if act_args is not None else None}
    return nn.Sequential(*[parts[k] for k in order.split('-') if parts[k] is not None])


def create_convblock2d(*args, norm_args=None, act_args=None, order='conv-norm-act', **kwargs):
    return _convblock(Conv2d, '2d', args, norm_args, act_args, order, kwargs)


def create_convblock1d(*args, norm_args=None, act_args=None, order='conv-norm-act', **kwargs):
    return _convblock(Conv1d, '1d', args, norm_args, act_args, order, kwargs)


def _fusable_bn(bn, x):
    """plain training-mode BatchNorm1d/2d on a contiguous fp32 GPU tensor (SyncBatchNorm, eval mode and
    other norms take the ordinary torch modules)"""
    return (type(bn) in (nn.BatchNorm1d, nn.BatchNorm2d) and bn.training and bn.affine and bn.track_running_stats
            and x.is_cuda and x.dtype == torch.float32)


def _eval_bn(bn, x):
    """inference-mode BatchNorm (any of the BatchNorm classes) on a contiguous fp32 GPU tensor outside autograd:
    the fused kernels with the running statistics (ops.bn_eval)"""
    return (isinstance(bn, nn.modules.batchnorm._BatchNorm) and not bn.training and bn.affine
            and bn.track_running_stats and bn.running_mean is not None and x.is_cuda and x.dtype == torch.float32
            and not torch.is_grad_enabled())


def _synced_bn_group(bn, x):
    """process group of a training-mode nn.SyncBatchNorm whose statistics really span several ranks (then the fused
    kernels exchange their per-channel sums: ops.SyncBatchNormFused), else None"""
    import torch.distributed as dist
    if not (type(bn) is nn.SyncBatchNorm and bn.training and bn.affine and bn.track_running_stats and x.is_cuda
            and x.dtype == torch.float32 and dist.is_available() and dist.is_initialized()):
        return None
    group = bn.process_group if bn.process_group is not None else dist.group.WORLD
    return group if dist.get_world_size(group) > 1 or _FORCE_SYNCED_BN else None


_FORCE_SYNCED_BN = False  # tests: take the cross-rank path in a one-rank group too


def conv1x1(conv, x):
    """conv(x); a plain 1x1 convolution on a contiguous fp32 GPU tensor runs on the MFMA kernels of
    csrc/pwconv.hip (same parameters, same autograd contract), anything else on the stored torch module."""
    if (type(conv) in (nn.Conv1d, nn.Conv2d, Conv1d, Conv2d) and x.is_cuda and x.dtype == torch.float32
            and conv.groups == 1 and conv.padding_mode == 'zeros'
            and all(k == 1 for k in conv.kernel_size) and all(v == 1 for v in conv.stride)
            and all(v == 0 for v in conv.padding) and x.dim() == conv.weight.dim()):
        from amcontrast3d_amd.ops import mixed_precision
        if mixed_precision() and _bf16_pays(conv, x):
            # use_amp (main_AA.py:389-394): bf16 operands, fp32 accumulation on the bf16 MFMA; activations, BatchNorm,
            # searches and the loss stay fp32 (tensors are never stored in bf16)
            from amcontrast3d_amd.ops import pointwise_conv
            return pointwise_conv(x, conv.weight, conv.bias, True)
        if _pw_pays(conv, x):
            from amcontrast3d_amd.ops import pointwise_conv
            return pointwise_conv(x, conv.weight, conv.bias)
        if (conv.bias is None and min(conv.in_channels, conv.out_channels) >= 64 and conv.in_channels % 4 == 0
                and torch.is_grad_enabled() and not os.environ.get("AMC3D_NO_LIBRARY_GEMM")):
            # deep and short (SA4, coarse FP stages): three plain library GEMMs instead of the convolution library,
            # whose weight gradient is wrapped in layout transposes (scratch/pw_bench3.py: 30-50 us per layer)
            from amcontrast3d_amd.ops import library_gemm_conv
            return library_gemm_conv(x, conv.weight)
        # what is left (small layers with a bias: the skip convs of SA2-4): this library's kernel too -- the convolution
        # library answers them with layout transposes around implicit-GEMM kernels (and a find-mode search at first use)
        if os.environ.get("AMC3D_SMALL_CONV_OWN"):  # measured: 0.35 ms/step slower than the convolution library on these three layers
            from amcontrast3d_amd.ops import pointwise_conv
            return pointwise_conv(x, conv.weight, conv.bias)
        with torch.autocast("cuda", enabled=False):
            return conv(x)
    return conv(x)


def conv1x1_weight(x, w):
    """1x1 convolution of x (B,Cin,P) with an explicit bias-free weight w (Cout,Cin): the routing of conv1x1 for a slice
    of a module's weight (the two halves of a FeaturePropogation conv)"""
    from amcontrast3d_amd import ops
    cin, cout = w.shape[1], w.shape[0]
    positions = x.numel() // x.shape[1]
    w3 = w.reshape(cout, cin, 1)
    if ops.mixed_precision() and min(cin, cout) >= 64 and positions >= _BF16_MIN_POSITIONS:
        return ops.pointwise_conv(x, w3, None, True)
    if (min(cin, cout) >= 64 and positions < 65536 and cin % 4 == 0 and torch.is_grad_enabled()
            and not os.environ.get("AMC3D_NO_LIBRARY_GEMM")):
        return ops.library_gemm_conv(x, w3)
    return ops.pointwise_conv(x, w3, None)


def feature_propagation_first_block(blk, f1, f2, geom):
    """First block of FeaturePropogation (pointnext_AA.py:210-226: interpolate f2 onto the fine cloud, concatenate with the
    skip features f1, Conv1d -> BatchNorm1d -> ReLU) with the conv applied BEFORE the interpolation:
        W . [f1 ; up(f2)] = W1 . f1 + up(W2 . f2)
    (the 3-NN interpolation is linear and mixes points, the 1x1 conv mixes channels: they commute).  The interpolated
    tensor has Cout instead of C2 channels (half at every level), its conv runs on the coarse cloud (4x fewer points), and
    no concatenated (B, C1+C2, n) tensor exists.  -> the block's output, or None when it is not of that form."""
    from amcontrast3d_amd import ops
    if (not isinstance(blk, nn.Sequential) or len(blk) != 3 or type(blk[0]) not in (nn.Conv1d, Conv1d)
            or not isinstance(blk[1], nn.modules.batchnorm._BatchNorm) or type(blk[2]) is not nn.ReLU
            or f1 is None or not f1.is_cuda or f1.dtype != torch.float32 or f2.dtype != torch.float32
            or os.environ.get("AMC3D_NO_FP_SPLIT")):
        return None
    conv, bn = blk[0], blk[1]
    c1 = f1.shape[1]
    if (conv.bias is not None or conv.kernel_size != (1,) or conv.stride != (1,) or conv.groups != 1
            or conv.in_channels != c1 + f2.shape[1]):
        return None
    w1, w2 = ops.split_weight(conv.weight, c1)
    y = ops.three_interpolate_add(conv1x1_weight(f2, w2), geom['idx'], geom['weight'], conv1x1_weight(f1, w1))
    group = _synced_bn_group(bn, y)
    if _eval_bn(bn, y):
        return ops.bn_eval(y, bn, True, False)
    if group is not None:
        return ops.SyncBatchNormFused.apply(y, bn.weight, bn.bias, bn.eps, True, False, bn, group)[0]
    if _fusable_bn(bn, y):
        return ops.BatchNormAct.apply(y, bn.weight, bn.bias, bn.eps, True, bn)[0]
    return blk[2](bn(y))


# (round 3, cfg 5 = XL + ++ at 1 x 120000: threshold 4096 -> 19.8 ms per step, 16384 -> 19.1, 65536 -> 19.4, never -> 19.3; the
# shorter deep layers take the library GEMMs, which run on bf16 operands under autocast: ops.LibraryGemmConv)
_BF16_MIN_POSITIONS = int(os.environ.get("AMC3D_BF16_MIN_POSITIONS", 16384))


def _bf16_pays(conv, x):
    """Under autocast the 1x1 convs with >= 64 channels on both sides over >= 16384 positions run on the bf16 MFMA
    (scratch/gemm_bench.py: 1.2-2x the fp32 kernels there).  Narrower layers are HBM-bound and shorter ones fill a fraction
    of the chip with 128 x 128 tiles: both keep their fp32 route, which is at least as accurate as what autocast asks for."""
    return min(conv.in_channels, conv.out_channels) >= 64 and x.numel() // x.shape[1] >= _BF16_MIN_POSITIONS


def _pw_pays(conv, x):
    """Measured on MI355X (scratch/pw_bench.py): the kernel beats MIOpen/rocBLAS (which wrap the weight gradient
    in NCHW<->NHWC transposes) on the wide, shallow layers -- stem, head, the two finest FeaturePropagation stages;
    the deep, narrow ones (>= 128 channels over <= 10^5 positions) are MFMA-bound GEMMs the library does better."""
    positions = x.numel() // x.shape[1]
    cin, cout = conv.in_channels, conv.out_channels
    if min(cin, cout) >= 64:  # csrc/gemm.hip: double-buffered 128x128 MFMA tiles, on par with the library GEMMs and
        return positions >= 65536  # without MIOpen's layout transposes around the weight gradient
    return positions >= 131072 and max(cin, cout) <= 128 or (max(cin, cout) <= 64 and positions >= 32768)


def run_convblocks(blocks, x, pool_max=False, pre=None, activated=False, residual=None):
    """Evaluate a stack of conv blocks (the nn.Sequential the factories above build), optionally followed by
    the max over the last (neighbour) dimension.  Where a block is conv -> plain BatchNorm [-> ReLU] in
    training mode, BatchNorm statistics, normalisation, ReLU and (for the last block) the max-pool run as
    fused gfx950 kernels (amcontrast3d_amd/csrc/bn.hip); everything else runs the stored modules as they are.
    Parameters, buffers and their bookkeeping stay those of the nn modules.
    `pre`: the already computed output of the first block's convolution (the fused gather+conv kernel).
    `activated`: x is the activated output of a first block evaluated elsewhere (fused_first_block); `blocks` are the rest.
    `residual`: the result is relu(stack(x) + residual) -- an InvResMLP block's `f += identity; act(f)` -- inside the last
    block's BatchNorm kernels where that block is conv -> plain training-mode BatchNorm without activation."""
    from amcontrast3d_amd.ops import BatchNormAct, BatchNormMax, BatchNormResidualAct, SyncBatchNormFused
    res_done = False
    mods = list(blocks)
    pooled = False
    fused = _sa_tail_activated(mods, x, pool_max) if activated else _sa_tail(mods, pool_max, pre)
    if fused is not None:
        return fused
    for bi, blk in enumerate(mods):
        last = bi == len(mods) - 1
        sub = list(blk) if isinstance(blk, nn.Sequential) else None
        if (sub is not None and len(sub) in (2, 3) and isinstance(sub[0], (nn.Conv1d, nn.Conv2d))
                and isinstance(sub[1], nn.modules.batchnorm._BatchNorm)
                and (len(sub) == 2 or type(sub[2]) is nn.ReLU)):
            y = pre if (bi == 0 and pre is not None) else conv1x1(sub[0], x)
            bn = sub[1]
            group = _synced_bn_group(bn, y)
            if _eval_bn(bn, y):
                from amcontrast3d_amd.ops import bn_eval
                pool = last and pool_max and y.dim() == 4 and y.shape[-1] <= 255
                x = bn_eval(y, bn, len(sub) == 3, pool)
                pooled = pooled or pool
            elif group is not None:
                pool = last and pool_max and y.dim() == 4 and y.shape[-1] <= 255
                x, _, _ = SyncBatchNormFused.apply(y, bn.weight, bn.bias, bn.eps, len(sub) == 3, pool, bn, group)
                pooled = pooled or pool
            elif _fusable_bn(bn, y):
                relu = len(sub) == 3
                # the kernels also do nn.BatchNorm's running-stat bookkeeping (same launch)
                if last and pool_max and y.dim() == 4 and y.shape[-1] <= 255:
                    x, _, _ = BatchNormMax.apply(y, bn.weight, bn.bias, bn.eps, relu, bn)
                    pooled = True
                elif (last and residual is not None and not relu and not pool_max and residual.shape == y.shape
                      and residual.is_cuda and residual.dtype == torch.float32 and not os.environ.get("AMC3D_NO_BN_RESIDUAL")):
                    x, _, _ = BatchNormResidualAct.apply(y, residual, bn.weight, bn.bias, bn.eps, bn)
                    res_done = True
                else:
                    x, _, _ = BatchNormAct.apply(y, bn.weight, bn.bias, bn.eps, relu, bn)
            else:
                x = bn(y)
                if len(sub) == 3:
                    x = sub[2](x)
        else:
            assert not (bi == 0 and pre is not None), "pre needs a conv -> norm block"
            if sub is not None and len(sub) >= 1 and isinstance(sub[0], (nn.Conv1d, nn.Conv2d)):
                x = conv1x1(sub[0], x)
                for mod in sub[1:]:
                    x = mod(x)
            else:
                x = blk(x)
    if pool_max and not pooled:
        x = torch.max(x, dim=-1, keepdim=False)[0]
    if residual is not None and not res_done:
        x = torch.relu(x + residual)
    return x


def _sa_tail(mods, pool_max, pre):
    """[conv0 (already applied: `pre`), BN, ReLU] -> [1x1 conv, BN (, ReLU)] -> max over 32 neighbours as one recomputing
    kernel family (csrc/sa_tail.hip), or None when the stack is not of that form."""
    import os
    if pre is None or not pool_max or len(mods) != 2 or os.environ.get("AMC3D_NO_SA_TAIL"):
        return None
    s0 = list(mods[0]) if isinstance(mods[0], nn.Sequential) else None
    s1 = list(mods[1]) if isinstance(mods[1], nn.Sequential) else None
    if (s0 is None or s1 is None or len(s0) != 3 or type(s0[2]) is not nn.ReLU or len(s1) not in (2, 3)
            or (len(s1) == 3 and type(s1[2]) is not nn.ReLU) or not isinstance(s1[0], nn.Conv2d)):
        return None
    bn1, conv2, bn2 = s0[1], s1[0], s1[1]
    if (pre.dim() != 4 or not _fusable_bn(bn1, pre) or not _fusable_bn(bn2, pre) or conv2.bias is not None
            or conv2.kernel_size != (1, 1) or conv2.stride != (1, 1) or conv2.groups != 1
            or any(v != 0 for v in conv2.padding) or conv2.in_channels != pre.shape[1]):
        return None
    from amcontrast3d_amd import ops
    if not (ops.sa_tail_supported(pre.shape[1], conv2.out_channels, pre.shape[-1])
            and ops.sa_tail_pays(pre.shape[1], conv2.out_channels)):
        return None
    return ops.SATail.apply(pre, bn1.weight, bn1.bias, bn1.eps, conv2.weight, bn2.weight, bn2.bias, bn2.eps,
                            len(s1) == 3, bn1, bn2)


def fused_local_aggregation(blocks, f, geom, feature_type):
    """A stack of ONE conv block -- Conv2d 1x1 -> BatchNorm2d [-> ReLU] -- followed by the max over the neighbours (every
    LocalAggregation of InvResMLP, the single-layer SetAbstraction of PointNeXt-B/L/XL) as convolve-before-gather
    (amcontrast3d_amd/csrc/lagg.hip): the pooled (B,C,M) output, or None when the layer is not of that form."""
    from amcontrast3d_amd import ops
    import os
    if (feature_type != 'dp_fj' or geom is None or 'idx' not in geom or geom.get('mom') is None or f is None or not f.is_cuda
            or f.dtype != torch.float32 or len(blocks) != 1
            or os.environ.get("AMC3D_NO_LOCAL_AGGREGATION")):
        return None
    blk = blocks[0]
    if (not isinstance(blk, nn.Sequential) or len(blk) not in (2, 3) or not isinstance(blk[0], nn.Conv2d)
            or not isinstance(blk[1], nn.modules.batchnorm._BatchNorm) or (len(blk) == 3 and type(blk[2]) is not nn.ReLU)):
        return None
    conv, bn = blk[0], blk[1]
    idx = geom['idx']
    if (conv.bias is not None or conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.groups != 1
            or any(v != 0 for v in conv.padding) or conv.in_channels != f.shape[1] + 3
            or not ops.local_aggregation_supported(conv.out_channels, idx.shape[-1])):
        return None
    relu = len(blk) == 3
    if _eval_bn(bn, f):
        return ops.local_aggregation_eval(f, geom['dp'], idx, conv.weight, bn, relu)
    group = _synced_bn_group(bn, f)  # nn.SyncBatchNorm over several ranks: the statistics are exchanged between two phases
    if group is None and not _fusable_bn(bn, f):
        return None
    return ops.LocalAggregationFused.apply(f, geom['dp'], idx, geom['mom'], conv.weight, bn.weight, bn.bias, bn.eps, relu, bn,
                                           group)


def fused_first_block(blocks, f, geom, feature_type):
    """First block -- Conv2d 1x1 -> BatchNorm2d -> ReLU -- of a multi-layer neighbourhood MLP (PointNeXt-S'
    SetAbstraction, sa_layers = 2) convolved before the gather, its ACTIVATED output x1 (B,C,M,32) materialised for the
    blocks that follow (ops.GroupedConvBN), or None when the stack is not of that form."""
    from amcontrast3d_amd import ops
    import os
    if (feature_type != 'dp_fj' or geom is None or 'idx' not in geom or geom.get('mom') is None or f is None or not f.is_cuda
            or f.dtype != torch.float32 or len(blocks) < 2 or os.environ.get("AMC3D_NO_LOCAL_AGGREGATION")):
        return None
    blk = blocks[0]
    if (not isinstance(blk, nn.Sequential) or len(blk) != 3 or not isinstance(blk[0], nn.Conv2d)
            or not isinstance(blk[1], nn.modules.batchnorm._BatchNorm) or type(blk[2]) is not nn.ReLU):
        return None
    conv, bn = blk[0], blk[1]
    idx = geom['idx']
    if (conv.bias is not None or conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.groups != 1
            or any(v != 0 for v in conv.padding) or conv.in_channels != f.shape[1] + 3
            or not ops.grouped_conv_bn_supported(conv.out_channels, idx.shape[-1])):
        return None
    if _eval_bn(bn, f):
        return ops.grouped_conv_bn_eval(f, geom['dp'], idx, conv.weight, bn, True)
    group = _synced_bn_group(bn, f)
    if group is None and not _fusable_bn(bn, f):
        return None
    csr = geom.get('csr')
    if csr is not None:
        csr = (csr['start'], csr['edge'], csr.get('edge_dp'))
    return ops.GroupedConvBN.apply(f, geom['dp'], idx, geom['mom'], conv.weight, bn.weight, bn.bias, bn.eps, True, bn, csr,
                                   group)


def _sa_tail_activated(mods, x1, pool_max):
    """[1x1 conv, BN (, ReLU)] -> max over 32 neighbours on the activated first-layer output x1, as the recomputing
    kernel family of csrc/sa_tail.hip (ops.SATailActivated), or None"""
    import os
    if not pool_max or len(mods) != 1 or x1.dim() != 4 or os.environ.get("AMC3D_NO_SA_TAIL"):
        return None
    s1 = list(mods[0]) if isinstance(mods[0], nn.Sequential) else None
    if (s1 is None or len(s1) not in (2, 3) or (len(s1) == 3 and type(s1[2]) is not nn.ReLU) or not isinstance(s1[0], nn.Conv2d)
            or not isinstance(s1[1], nn.modules.batchnorm._BatchNorm)):
        return None
    conv2, bn2 = s1[0], s1[1]
    if (not _fusable_bn(bn2, x1) or conv2.bias is not None or conv2.kernel_size != (1, 1) or conv2.stride != (1, 1)
            or conv2.groups != 1 or any(v != 0 for v in conv2.padding) or conv2.in_channels != x1.shape[1]):
        return None
    from amcontrast3d_amd import ops
    if not (ops.sa_tail_supported(x1.shape[1], conv2.out_channels, x1.shape[-1])
            and ops.sa_tail_pays(x1.shape[1], conv2.out_channels)):
        return None
    return ops.SATailActivated.apply(x1, conv2.weight, bn2.weight, bn2.bias, bn2.eps, len(s1) == 3, bn2)


def fused_first_conv(blocks, f, geom, feature_type):
    """Output of the first block's 1x1 conv on [dp ; f[idx]] from the fused gather+conv MFMA kernel, or None
    when the layer is not of that form (then the caller groups, concatenates and convolves as usual)."""
    from amcontrast3d_amd import ops
    blk = blocks[0]
    if (feature_type != 'dp_fj' or geom is None or 'idx' not in geom or f is None or not f.is_cuda
            or f.dtype != torch.float32 or not isinstance(blk, nn.Sequential)
            or len(blk) < 2 or not isinstance(blk[0], nn.Conv2d)
            or not isinstance(blk[1], nn.modules.batchnorm._BatchNorm)):
        return None
    conv = blk[0]
    if (conv.bias is not None or conv.kernel_size != (1, 1) or conv.stride != (1, 1) or conv.groups != 1
            or conv.in_channels != f.shape[1] + 3 or not ops.grouped_conv_supported(f.shape[1], conv.out_channels)):
        return None
    return ops.grouped_conv(f, geom['dp'], geom['idx'], conv.weight)


# input width of the first grouped conv for each neighbourhood feature recipe
# (models/layers/local_aggregation.py:13-29)
CHANNEL_MAP = {
    'fj': lambda x: x, 'df': lambda x: x, 'assa': lambda x: x * 3, 'assa_dp': lambda x: x * 3 + 3,
    'dp_fj': lambda x: 3 + x, 'pj': lambda x: x, 'dp': lambda x: 3, 'pi_dp': lambda x: x + 3,
    'pj_dp': lambda x: x + 3, 'dp_fj_df': lambda x: x * 2 + 3, 'dp_fi_df': lambda x: x * 2 + 3,
    'pi_dp_fj_df': lambda x: x * 2 + 6, 'pj_dp_fj_df': lambda x: x * 2 + 6, 'pj_dp_df': lambda x: x + 6,
    'dp_df': lambda x: x + 3,
}
