"""Image decoding for the dataset classes (PIL only; torchvision and imageio are not required)."""
from __future__ import annotations

import numpy as np
import torch


def _pil():
    try:
        from PIL import Image
    except ImportError as e:  # pragma: no cover
        raise ImportError("reading image files needs Pillow; pass decoded arrays instead") from e
    return Image


def load_rgb_resized(path, width: int, height: int) -> torch.Tensor:
    """File -> float [3, height, width] in [0, 1]: convert('RGB'), bilinear PIL resize, /255
    (what T.Resize + T.ToTensor do to a PIL image, src/data/nerf_capture_dataset.py:68-71, 88-92)."""
    Image = _pil()
    img = Image.open(path).convert("RGB").resize((width, height), Image.BILINEAR)
    return torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255.0)


def load_array(path) -> np.ndarray:
    """File -> the decoded array as stored (uint8 [H, W, C]), like imageio.v2.imread."""
    Image = _pil()
    return np.asarray(Image.open(path))
