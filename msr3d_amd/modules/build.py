"""`modules.build` of the reference (/root/reference/modules/build.py:6-21): four
registries and `build_module(type, cfg)` = REG.get(cfg.name)(cfg, **cfg.args)."""
from ..config import cfg2dict
from ..registry import Registry

VISION_REGISTRY = Registry("vision")
LANGUAGE_REGISTRY = Registry("language")
GROUNDING_REGISTRY = Registry("grounding")
HEADS_REGISTRY = Registry("heads")

_BY_TYPE = {
    "vision": VISION_REGISTRY,
    "language": LANGUAGE_REGISTRY,
    "grounding": GROUNDING_REGISTRY,
    "heads": HEADS_REGISTRY,
}


def build_module(module_type, cfg):
    if module_type not in _BY_TYPE:
        raise NotImplementedError(f"module type {module_type} not implemented")
    return _BY_TYPE[module_type].get(cfg.name)(cfg, **cfg2dict(cfg.args))
