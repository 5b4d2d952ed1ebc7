"""Optimizer construction of the trainer (openpoints/optim/optim_factory.py:66-230): the no-decay parameter grouping and
``build_optimizer_from_cfg(model, **cfg.optimizer)``.

What the shipped configs use -- ``NAME: 'adamw'``, lr 0.01, weight_decay 1e-4 (cfgs/s3dis/default.yaml:64-69) -- and the
plain torch optimizers (sgd / nesterov / momentum / adam / adamw) are built here.  On the GPU 'adamw' is this library's
``FusedAdamW`` (amcontrast3d_amd/fused_optim.py: same arguments, groups and state_dict as torch.optim.AdamW, two launches per step,
``step(max_grad_norm=...)`` folds clip_grad_norm_ in; AMC3D_TORCH_ADAMW=1 keeps torch's), Adam is created
``fused=True, capturable=True`` so that the update is a few multi-tensor launches that a hipGraph can replay.  The reference's
collection of third-party optimizers (AdaBelief, Lamb, MADGRAD, ...) is outside the hot path: ask for one and the error
names it (with AMC3D_REFERENCE_ROOT set, ``openpoints.optim`` of the reference tree can be imported instead).
"""
import json
import logging

import torch
import torch.nn as nn
import torch.optim as optim


def get_parameter_groups(model, weight_decay=1e-5, skip_list=(), get_num_layer=None, get_layer_scale=None,
                         filter_by_modules_names=None):
    """1-d parameters (BatchNorm scale / shift), '.bias' and names containing a skip_list entry -> weight decay 0
    (optim_factory.py:66-120); groups keep the reference's names and 'lr_scale' field"""
    names, groups = {}, {}
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        if len(param.shape) == 1 or name.endswith(".bias") or any(key in name for key in skip_list):
            group_name, this_decay = "no_decay", 0.
        else:
            group_name, this_decay = "decay", weight_decay
        layer_id = get_num_layer(name) if get_num_layer is not None else None
        if layer_id is not None:
            group_name = "layer_%d_%s" % (layer_id, group_name)
        scale = get_layer_scale(layer_id) if get_layer_scale is not None else 1.0
        if filter_by_modules_names is not None:
            for module_name, opts in filter_by_modules_names.items():
                if module_name in name:
                    group_name = module_name + '_' + group_name
                    this_decay = opts.get('weight_decay', this_decay)
                    scale = opts.get('lr_scale', 1.0) * scale
                    break
        if group_name not in groups:
            names[group_name] = {"weight_decay": this_decay, "params": [], "lr_scale": scale}
            groups[group_name] = {"weight_decay": this_decay, "params": [], "lr_scale": scale}
        groups[group_name]["params"].append(param)
        names[group_name]["params"].append(name)
    logging.info("Param groups = %s" % json.dumps(names, indent=2))
    return list(groups.values())


def add_weight_decay(model, weight_decay=1e-5, skip_list=()):
    decay, no_decay = [], []
    for name, param in model.named_parameters():
        if not param.requires_grad:
            continue
        (no_decay if (len(param.shape) == 1 or name.endswith(".bias") or name in skip_list) else decay).append(param)
    return [{'params': no_decay, 'weight_decay': 0.}, {'params': decay, 'weight_decay': weight_decay}]


def optimizer_kwargs(cfg):
    kwargs = dict(opt=cfg.opt, lr=cfg.lr, weight_decay=cfg.weight_decay, momentum=cfg.momentum)
    if getattr(cfg, 'opt_eps', None) is not None:
        kwargs['eps'] = cfg.opt_eps
    if getattr(cfg, 'opt_betas', None) is not None:
        kwargs['betas'] = cfg.opt_betas
    if getattr(cfg, 'opt_args', None) is not None:
        kwargs.update(cfg.opt_args)
    return kwargs


def build_optimizer_from_cfg(model, NAME='sgd', lr=None, weight_decay=0., momentum=0.9, filter_bias_and_bn=True,
                             filter_by_modules_names=None, **kwargs):
    assert isinstance(model, nn.Module)
    if 0. < kwargs.get('layer_decay', 0) < 1.0:
        raise NotImplementedError("layer-wise lr decay is a ViT option of the reference, outside the PointNeXt path")
    kwargs.pop('layer_decay', None)
    if weight_decay and filter_bias_and_bn:
        inner = model.module if hasattr(model, 'module') else model
        skip = inner.no_weight_decay() if hasattr(inner, 'no_weight_decay') else {}
        parameters = get_parameter_groups(model, weight_decay, skip, None, None, filter_by_modules_names)
        weight_decay = 0.
    else:
        parameters = list(model.parameters())
    opt = NAME.lower().split('_')[-1]
    args = dict(weight_decay=weight_decay, **kwargs)
    if lr is not None:
        args.setdefault('lr', lr)
    on_gpu = any(p.is_cuda for p in model.parameters())
    if opt in ('sgd', 'nesterov'):
        args.pop('eps', None)
        return optim.SGD(parameters, momentum=momentum, nesterov=True, **args)
    if opt == 'momentum':
        args.pop('eps', None)
        return optim.SGD(parameters, momentum=momentum, nesterov=False, **args)
    if opt in ('adam', 'adamw'):
        import os
        plist = [p for g in parameters for p in g['params']] if parameters and isinstance(parameters[0], dict) else parameters
        if (opt == 'adamw' and on_gpu and not os.environ.get("AMC3D_TORCH_ADAMW") and set(args) <= {'lr', 'weight_decay', 'betas', 'eps'}
                and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in plist)):
            # this library's AdamW: one table-driven launch over all parameter tensors, gradient-norm clipping folded into
            # step(max_grad_norm=...) (amcontrast3d_amd/fused_optim.py); a torch.optim.Optimizer with torch's state_dict layout
            from amcontrast3d_amd.fused_optim import FusedAdamW
            return FusedAdamW(parameters, **args)
        if on_gpu:
            args.setdefault('fused', True)
            args.setdefault('capturable', True)
        return (optim.AdamW if opt == 'adamw' else optim.Adam)(parameters, **args)
    raise NotImplementedError(f"optimizer {NAME!r}: only sgd / nesterov / momentum / adam / adamw are built here "
                              f"(the AMContrast3D configs use adamw)")
