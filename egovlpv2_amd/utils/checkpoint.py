"""Checkpoint format of the reference trainer (base/base_trainer.py:412-495): one ``torch.save`` dict with the keys
'arch', 'epoch', 'state_dict', 'optimizer', 'scheduler', 'monitor_best', 'config', so that checkpoints written by either
code base load in the other.  Parameter names are the reference's; FusedAdamW keeps the HF AdamW state
keys ('step', 'exp_avg', 'exp_avg_sq'), so optimiser state round-trips as well."""
from collections import OrderedDict

import torch


def _unwrap(model):
    return model.module if hasattr(model, 'module') else model


def save_checkpoint(path, model, optimizer, scheduler, epoch, monitor_best, config, save_best_path=None):
    """_save_checkpoint (:412-436).  `model` may be DDP-wrapped: like the reference, the keys are then 'module.'-prefixed."""
    state = {
        'arch': type(_unwrap(model)).__name__ if not hasattr(model, 'module') else type(model).__name__,
        'epoch': epoch,
        'state_dict': model.state_dict(),
        'optimizer': optimizer.state_dict() if optimizer is not None else None,
        'scheduler': scheduler.state_dict() if scheduler is not None else None,
        'monitor_best': monitor_best,
        'config': config,
    }
    torch.save(state, path)
    if save_best_path is not None:
        torch.save(state, save_best_path)
    return state


def match_data_parallel_keys(state_dict, model_keys):
    """add or strip the 'module.' prefix so that `state_dict` matches the receiving model (:456-481)"""
    load_keys = list(state_dict.keys())
    if not load_keys or not model_keys:
        return state_dict
    cur_dp, load_dp = model_keys[0].startswith('module.'), load_keys[0].startswith('module.')
    if load_dp and not cur_dp:
        return OrderedDict((k[7:], v) for k, v in state_dict.items())
    if cur_dp and not load_dp:
        return OrderedDict(('module.' + k, v) for k, v in state_dict.items())
    return state_dict


def resume_checkpoint(path, model, optimizer=None, scheduler=None, config=None, map_location='cpu', logger=None):
    """_resume_checkpoint (:438-495).  Returns (start_epoch, monitor_best).  Optimiser and scheduler state are restored
    only when the optimiser type in the checkpoint's config equals the current one (same rule as the reference)."""
    ck = torch.load(path, map_location=map_location, weights_only=False)
    start_epoch = ck['epoch'] + 1
    if config is not None and ck.get('config') is not None and ck['config'].get('arch') != config.get('arch') and logger:
        logger.warning("Architecture configuration given in config file is different from that of checkpoint.")
    sd = match_data_parallel_keys(ck['state_dict'], list(model.state_dict().keys()))
    model.load_state_dict(sd)
    same_opt = True
    if config is not None and ck.get('config') is not None:
        same_opt = ck['config'].get('optimizer', {}).get('type') == config.get('optimizer', {}).get('type')
    if same_opt:
        if optimizer is not None and ck.get('optimizer') is not None:
            optimizer.load_state_dict(ck['optimizer'])
        if scheduler is not None and ck.get('scheduler') is not None:
            scheduler.load_state_dict(ck['scheduler'])
    elif logger:
        logger.warning("Optimizer type given in config file is different from that of checkpoint. "
                       "Optimizer parameters not being resumed.")
    return start_epoch, ck['monitor_best']
