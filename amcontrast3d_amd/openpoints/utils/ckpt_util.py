"""Checkpoint helpers of the trainer, under the reference's names (openpoints/utils/ckpt_util.py:12-183).

File layout is the reference's: ``{'model', 'optimizer', 'scheduler', 'epoch', **additional}`` saved as
``<ckpt_dir>/<run_name>_<post_fix>.pth`` (+ ``_E<epoch>`` milestones every ``save_freq`` epochs, + ``_ckpt_best``), so
checkpoints written by either code base load in the other: the model's state-dict keys are identical
(tests/golden/state_keys.json).  Files are read with ``torch.load(..., weights_only=True)``: a checkpoint is tensors,
numbers, strings and containers -- nothing in it needs to execute.
"""
import logging
import os
import shutil
from collections import OrderedDict

import torch


def cal_model_parm_nums(model):
    return sum(p.nelement() for p in model.parameters())


def _load(path):
    return torch.load(path, map_location='cpu', weights_only=True)


def _latest(cfg, pretrained_path):
    if pretrained_path is not None:
        return pretrained_path
    return os.path.join(cfg.ckpt_dir, os.path.join(cfg.run_name, '_ckpt_latest.pth'))


def resume_model(model, cfg, pretrained_path=None):
    """strict load of ['model'] (a DataParallel 'module.' prefix is dropped) -> (start_epoch, best_metrics); (0, 0) when
    the file does not exist (ckpt_util.py:18-45)"""
    path = _latest(cfg, pretrained_path)
    if not os.path.exists(path):
        logging.info(f'[RESUME INFO] no checkpoint file from path {path}...')
        return 0, 0
    state = _load(path)
    model.load_state_dict({k.replace("module.", ""): v for k, v in state['model'].items()}, strict=True)
    start_epoch = state['epoch'] + 1 if 'epoch' in state else 1
    best = state.get('best_metrics')
    if best is not None and not isinstance(best, dict):
        best = best.state_dict()
    logging.info(f'[RESUME INFO] resume ckpts @ {start_epoch - 1} epoch( best_metrics = {best!s})')
    return start_epoch, best


def resume_optimizer(cfg, optimizer, pretrained_path=None):
    path = _latest(cfg, pretrained_path)
    if not os.path.exists(path):
        logging.info(f'[RESUME INFO] no checkpoint file from path {path}...')
        return 0, 0, 0
    state = _load(path)
    if state.get('optimizer'):
        optimizer.load_state_dict(state['optimizer'])


def save_checkpoint(cfg, model, epoch, optimizer=None, scheduler=None, additioanl_dict=None, is_best=False,
                    post_fix='ckpt_latest', save_name=None):
    """(the keyword `additioanl_dict` is spelled as the reference's callers spell it, ckpt_util.py:61-63)"""
    if save_name is None:
        save_name = cfg.run_name
    path = os.path.join(cfg.ckpt_dir, f'{save_name}_{post_fix}.pth')
    save_dict = {
        'model': model.module.state_dict() if hasattr(model, 'module') else model.state_dict(),
        'optimizer': optimizer.state_dict() if optimizer is not None else dict(),
        'scheduler': scheduler.state_dict() if scheduler is not None else dict(),
        'epoch': epoch,
    }
    if additioanl_dict is not None:
        save_dict.update(additioanl_dict)
    torch.save(save_dict, path)
    if cfg.save_freq > 0 and epoch % cfg.save_freq == 0:
        milestone = os.path.join(cfg.ckpt_dir, f'{save_name}_E{epoch}.pth')
        shutil.copyfile(path, milestone)
        logging.info("Saved in {}".format(milestone))
    if is_best:
        best = os.path.join(cfg.ckpt_dir, f'{save_name}_ckpt_best.pth' if save_name else 'ckpt_best.pth')
        shutil.copyfile(path, best)
        logging.info("Found the best model and saved in {}".format(best))


def resume_checkpoint(config, model, optimizer=None, scheduler=None, pretrained_path=None, printer=logging.info):
    """model + optimizer + scheduler + epoch counters (ckpt_util.py:93-133); a 'module.' prefix is added or removed to
    match the model; an optimizer / scheduler state that does not fit is reported, not fatal"""
    if pretrained_path is None:
        pretrained_path = config.pretrained_path
        assert pretrained_path is not None
    printer("=> loading checkpoint '{}'".format(pretrained_path))
    checkpoint = _load(pretrained_path)
    if optimizer is not None:
        try:
            optimizer.load_state_dict(checkpoint['optimizer'])
        except Exception:
            printer('optimizer does not match')
    if scheduler is not None:
        try:
            scheduler.load_state_dict(checkpoint['scheduler'])
        except Exception:
            printer('scheduler does not match')
    ckpt_state = checkpoint['model']
    model_multi = list(model.state_dict())[0].split('.')[0] == 'module'
    ckpt_multi = list(ckpt_state)[0].split('.')[0] == 'module'
    if model_multi != ckpt_multi:
        ckpt_state = OrderedDict((k[7:] if ckpt_multi else 'module.' + k, v) for k, v in ckpt_state.items())
    model.load_state_dict(ckpt_state)
    config.start_epoch = checkpoint['epoch'] + 1
    config.epoch = checkpoint['epoch'] + 1
    printer("=> loaded successfully '{}' (epoch {})".format(pretrained_path, checkpoint['epoch']))


def load_checkpoint(model, pretrained_path, module=None):
    """non-strict load for testing / fine-tuning (ckpt_util.py:136-183) -> (epoch, metrics found in the file)"""
    if not os.path.exists(pretrained_path):
        raise NotImplementedError('no checkpoint file from path %s...' % pretrained_path)
    state = _load(pretrained_path)
    ckpt = state
    for key in state.keys():
        if key in ['model', 'net', 'network', 'state_dict', 'base_model']:
            ckpt = ckpt[key]
    base = {k.replace("module.", ""): v for k, v in ckpt.items()}
    if module is not None:
        base = {k: v for k, v in base.items() if module in k}
    target = model.module if hasattr(model, 'module') else model
    incompatible = target.load_state_dict(base, strict=False)
    if incompatible.missing_keys:
        logging.info(get_missing_parameters_message(incompatible.missing_keys))
    if incompatible.unexpected_keys:
        logging.info(get_unexpected_parameters_message(incompatible.unexpected_keys))
    logging.info(f'Successful Loading the ckpt from {pretrained_path}')
    epoch = state.get('epoch', -1)
    metrics = {k: v for k, v in state.items() if any(t in k for t in ('metric', 'acc', 'test', 'val'))}
    logging.info(f'ckpts @ {epoch} epoch( {metrics} )')
    return epoch, metrics


def _grouped(keys):
    groups = {}
    for k in keys:
        head, _, tail = k.rpartition('.')
        groups.setdefault(head, []).append(tail)
    return "\n".join("  " + (h + '.' if h else '') + ('{' + ', '.join(t) + '}' if len(t) > 1 else t[0]) for h, t in groups.items())


def get_missing_parameters_message(keys):
    return "Some model parameters or buffers are not found in the checkpoint:\n" + _grouped(keys)


def get_unexpected_parameters_message(keys):
    return "The checkpoint state_dict contains keys that are not used by the model:\n" + _grouped(keys)
