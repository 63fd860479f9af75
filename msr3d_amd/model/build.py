"""Model registry of the hot path.

Same surface as the reference's `model.build` (/root/reference/model/build.py:6-19) --
`MODEL_REGISTRY`, `BaseModel`, `build_model(cfg)` looking the class up by `cfg.model.name`
and constructing it with the whole config -- on the package's own `Registry` instead of
fvcore's.
"""
from torch import nn

from ..registry import Registry

MODEL_REGISTRY = Registry("model")


def build_model(cfg):
    """Instantiate `MODEL_REGISTRY[cfg.model.name]` with `cfg` (the constructor reads what it
    needs from it, e.g. OSE3DSituation takes `cfg.model.*`)."""
    name = cfg.model.name
    if name not in MODEL_REGISTRY:
        known = ", ".join(k for k, _ in MODEL_REGISTRY)
        raise KeyError(f"unknown model '{name}' (registered: {known})")
    cls = MODEL_REGISTRY.get(name)
    return cls(cfg)


class BaseModel(nn.Module):
    """Base class of registry models; subclasses say which parameters the optimiser gets."""

    def __init__(self, cfg):
        super().__init__()

    def get_opt_params(self):
        raise NotImplementedError("Function to obtain all default parameters for optimization")
