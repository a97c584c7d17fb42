"""state_dict_data_parallel_fix (reference utils/util.py:31-57): add/strip the ``module.`` prefix so that
checkpoints saved from a DDP-wrapped model load into a bare one and vice versa."""
from collections import OrderedDict


def state_dict_data_parallel_fix(load_state_dict, curr_state_dict):
    load_keys, curr_keys = list(load_state_dict.keys()), list(curr_state_dict.keys())
    if not load_keys or not curr_keys:
        return load_state_dict
    load_dp, curr_dp = load_keys[0].startswith('module.'), curr_keys[0].startswith('module.')
    if load_dp and not curr_dp:
        return OrderedDict((k[len('module.'):], v) for k, v in load_state_dict.items())
    if curr_dp and not load_dp:
        return OrderedDict(('module.' + k, v) for k, v in load_state_dict.items())
    return load_state_dict
