"""Checkpoint format of the reference trainer (base/base_trainer.py:412-495): one ``torch.save`` dict with the keys
'arch', 'epoch', 'state_dict', 'optimizer', 'scheduler', 'monitor_best', 'config', so that checkpoints written by either
code base load in the other.  Parameter names are the reference's; FusedAdamW keeps the HF AdamW state
keys ('step', 'exp_avg', 'exp_avg_sq'), so optimiser state round-trips as well."""
from collections import OrderedDict

import torch

from .. import switches as SW


class _InertObject:
    """what a non-tensor object pickled inside a checkpoint becomes when the file is read without executing its code: a bag of
    the pickled attributes.  Subscripting falls through to a `config` / `_config` mapping when the object carried one (the
    reference's ConfigParser does: base_trainer.py:486 reads checkpoint['config']['optimizer']['type'])."""

    def __init__(self, *args, **kwargs):
        self._pickled_args = args

    def _mapping(self):
        for k in ('_config', 'config'):
            m = self.__dict__.get(k)
            if isinstance(m, dict):
                return m
        return {}

    def __getitem__(self, key):
        return self._mapping()[key]

    def get(self, key, default=None):
        return self._mapping().get(key, default)


def load_checkpoint_file(path, map_location='cpu', allow_unsafe=None):
    """torch.load without arbitrary code execution.  Reference checkpoints pickle a ConfigParser object under 'config'
    (base_trainer.py:412-436), which torch's weights-only unpickler refuses by name; plain `weights_only=False` would run whatever
    a checkpoint file says.  Here the file is first read weights-only; if it names classes outside torch's allow-list, each of
    them is replaced by an inert attribute bag of the same qualified name for the duration of the load (still the weights-only
    unpickler: no imports, no foreign constructors, no __reduce__ callables).  Only allow_unsafe=True, or EGV_ALLOW_UNSAFE_CHECKPOINT=1,
    falls back to the unrestricted pickle load for files even that cannot read."""
    import os
    import pickle
    try:
        return torch.load(path, map_location=map_location, weights_only=True)
    except pickle.UnpicklingError as first:
        err = first
    try:
        names = torch.serialization.get_unsafe_globals_in_checkpoint(path)
        stubs = [type(n.rsplit('.', 1)[-1], (_InertObject,), {'__module__': n.rsplit('.', 1)[0] if '.' in n else 'builtins'}) for n in names]
        with torch.serialization.safe_globals(stubs):
            return torch.load(path, map_location=map_location, weights_only=True)
    except Exception as second:                          # noqa: BLE001 -- report both, decide below
        err = second
    if allow_unsafe or (allow_unsafe is None and SW.on('EGV_ALLOW_UNSAFE_CHECKPOINT')):
        return torch.load(path, map_location=map_location, weights_only=False)
    raise RuntimeError(f"{path}: cannot be read without executing pickled code ({type(err).__name__}: {err}); if the file is trusted, "
                       f"pass allow_unsafe=True or set EGV_ALLOW_UNSAFE_CHECKPOINT=1") from err


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
    ck = load_checkpoint_file(path, map_location=map_location)
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
