"""Registry-built models of the hot path (mirror of the reference's `model/` package)."""
from .build import MODEL_REGISTRY, BaseModel, build_model  # noqa: F401
from . import ose3d_situation  # noqa: F401  (registers OSE3DSituation)
from . import scene_embeds  # noqa: F401  (registers MSR3DHotPath)
from . import msr3d_full  # noqa: F401  (registers MSR3DFullStep)
