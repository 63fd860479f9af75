"""Registry-built modules of the hot path (mirror of the reference's `modules/`
package, restricted to what OSE3DSituation instantiates)."""
from .build import (GROUNDING_REGISTRY, HEADS_REGISTRY, LANGUAGE_REGISTRY,  # noqa: F401
                    VISION_REGISTRY, build_module)
from .vision import pcd_pointnet_encoder  # noqa: F401  (registers PcdObjEncoder)
