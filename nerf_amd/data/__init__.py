"""Dataset formats of the reference (src/data/) and a GPU-resident ray buffer over them."""
from .nerf_capture_dataset import NeRFCaptureDataset, normalize_poses, read_transforms  # noqa: F401
from .LLFFDataset import LLFFDataset, read_poses_bounds  # noqa: F401
from .ray_buffer import RayBuffer  # noqa: F401
