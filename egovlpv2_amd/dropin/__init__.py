"""Drop-in registration of the HIP hot path under the REFERENCE's module names.

The reference's launcher binds its model code by module name (multinode_train_egoclip.py:23-26):

    import model.metric as module_metric
    import model.loss as module_loss
    import model.model as module_arch

and instantiates `getattr(module_arch, config['arch']['type'])`.  With ONE added line ahead of those imports,

    import egovlpv2_amd.dropin; egovlpv2_amd.dropin.install()

the names `model.model`, `model.loss` (and `model.model_epic_charades` for the fine-tune launchers) resolve to this package's
modules -- same class names, constructor and forward() signatures, state-dict keys -- while every other module of the reference tree
(`model` the package itself, `data_loader`, `trainer`, `parse_config`, `logger`, `utils` ...) is imported from the reference as before:
nothing else of the tree is shadowed.  The hot-path helpers that live in the reference's trainer / utils modules are patched in
place when those modules load, only where the replacement is value-identical:

    trainer.trainer_egoclip.AllGather_multi      (trainer/trainer_egoclip.py:25-41)
    utils.util.state_dict_data_parallel_fix      (utils/util.py:31-53)
    set_optim_schedule.set_schedule              (set_optim_schedule.py:16-129; opt-in: install(optimizer=True))

install() is idempotent; uninstall() restores sys.modules / sys.meta_path.
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

_ALIASES = {
    'model.model': 'egovlpv2_amd.model.model',
    'model.loss': 'egovlpv2_amd.model.loss',
    'model.model_epic_charades': 'egovlpv2_amd.model.model_epic_charades',
}
_PATCHES = {
    'trainer.trainer_egoclip': (('AllGather_multi', 'egovlpv2_amd.trainer.trainer_egoclip'),),
    'utils.util': (('state_dict_data_parallel_fix', 'egovlpv2_amd.utils.util'),),
}
_OPT_PATCHES = {
    'set_optim_schedule': (('set_schedule', 'egovlpv2_amd.set_optim_schedule'),),
}


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        return importlib.import_module(self.target)

    def exec_module(self, module):            # the target module is already initialised
        pass


class _Finder(importlib.abc.MetaPathFinder):
    """serves the aliased names; lets everything else through and patches the listed attributes after the real import"""

    def __init__(self, patches):
        self.patches = patches
        self._busy = set()

    def find_spec(self, fullname, path=None, target=None):
        tgt = _ALIASES.get(fullname)
        if tgt is not None:
            return importlib.machinery.ModuleSpec(fullname, _AliasLoader(tgt))
        if fullname in self.patches and fullname not in self._busy:
            self._busy.add(fullname)
            try:
                spec = importlib.util.find_spec(fullname)
            finally:
                self._busy.discard(fullname)
            if spec is None or spec.loader is None:
                return None
            spec.loader = _PatchingLoader(spec.loader, self.patches[fullname])
            return spec
        return None


class _PatchingLoader(importlib.abc.Loader):
    def __init__(self, inner, patches):
        self.inner, self.patches = inner, patches

    def create_module(self, spec):
        return self.inner.create_module(spec)

    def exec_module(self, module):
        self.inner.exec_module(module)
        for attr, src in self.patches:
            setattr(module, attr, getattr(importlib.import_module(src), attr))


_installed = []


def install(optimizer: bool = False):
    """register the aliases (see the module docstring).  optimizer=True also replaces set_optim_schedule.set_schedule by the fused
    HIP AdamW + schedule of this package."""
    import importlib.util  # noqa: F401  (find_spec above)
    if _installed:
        return
    patches = dict(_PATCHES)
    if optimizer:
        patches.update(_OPT_PATCHES)
    f = _Finder(patches)
    sys.meta_path.insert(0, f)
    _installed.append(f)
    # modules of the reference that were imported BEFORE install(): alias / patch them now
    for name, tgt in _ALIASES.items():
        if name in sys.modules:
            sys.modules[name] = importlib.import_module(tgt)
    for name, plist in patches.items():
        mod = sys.modules.get(name)
        if mod is not None:
            for attr, src in plist:
                setattr(mod, attr, getattr(importlib.import_module(src), attr))


def uninstall():
    while _installed:
        f = _installed.pop()
        if f in sys.meta_path:
            sys.meta_path.remove(f)
    for name, tgt in _ALIASES.items():
        m = sys.modules.get(name)
        if m is not None and m.__name__ == tgt:
            del sys.modules[name]
