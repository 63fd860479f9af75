"""Device-side construction of the hot path's inputs: scans resident in
HBM, one launch per batch instead of the reference's host-side numpy chain."""
from .scene_store import SceneStore, load_scan_pth  # noqa: F401
from .scene_input import SceneInputBuilder, build_rotate_mat, rotate_situation  # noqa: F401
