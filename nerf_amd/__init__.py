"""nerf_amd: the NeRF render/train hot path of skowae/NeRF on MI355X (gfx950).

Same Python surface as the reference's ``src/`` (Camera, models.*, utils.*),
backed by libnerf_hip.so.  Importing the package does not load the library;
the first op does, and raises if it has not been built.
"""
from .Camera import Camera, look_at_camera  # noqa: F401

__all__ = ["Camera", "look_at_camera"]
