"""Name -> class registries: the reference's plug-in API (dassl/utils/registry.py:18-80, dassl/engine/build.py:3-11,
dassl/evaluation/build.py:3-11).  ``@TRAINER_REGISTRY.register()`` on a class, ``build_trainer(cfg)`` to look it up."""
from __future__ import annotations


class Registry:
    def __init__(self, name: str):
        self._name = name
        self._obj_map = {}

    def _do_register(self, name, obj, force=False):
        if name in self._obj_map and not force:
            raise KeyError(f'An object named "{name}" was already registered in "{self._name}" registry')
        self._obj_map[name] = obj

    def register(self, obj=None, force=False):
        if obj is None:
            def wrapper(fn_or_class):
                self._do_register(fn_or_class.__name__, fn_or_class, force=force)
                return fn_or_class
            return wrapper
        self._do_register(obj.__name__, obj, force=force)
        return obj

    def get(self, name):
        if name not in self._obj_map:
            raise KeyError(f'Object name "{name}" does not exist in "{self._name}" registry')
        return self._obj_map[name]

    def registered_names(self):
        return list(self._obj_map.keys())


TRAINER_REGISTRY = Registry("TRAINER")
EVALUATOR_REGISTRY = Registry("EVALUATOR")


def build_trainer(cfg, **kwargs):
    from . import trainers  # noqa: F401  (registers the plug-ins)
    avail = TRAINER_REGISTRY.registered_names()
    if cfg.TRAINER.NAME not in avail:
        raise ValueError(f"trainer {cfg.TRAINER.NAME} not in {avail}")
    return TRAINER_REGISTRY.get(cfg.TRAINER.NAME)(cfg, **kwargs)


def build_evaluator(cfg, **kwargs):
    from . import evaluation  # noqa: F401
    avail = EVALUATOR_REGISTRY.registered_names()
    if cfg.TEST.EVALUATOR not in avail:
        raise ValueError(f"evaluator {cfg.TEST.EVALUATOR} not in {avail}")
    return EVALUATOR_REGISTRY.get(cfg.TEST.EVALUATOR)(cfg, **kwargs)
