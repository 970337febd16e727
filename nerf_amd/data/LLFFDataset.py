"""LLFF scenes: ``poses_bounds.npy`` + ``images_<factor>/`` (src/data/LLFFDataset.py)."""
from __future__ import annotations

from pathlib import Path
from typing import List, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

from ._images import load_array

IMAGE_SUFFIXES = ("png", "jpg", "PNG", "JPG")   # the order the reference concatenates its globs in
TARGET_WIDTH = 100                               # images wider than this are shrunk to it (src/data/LLFFDataset.py:62)


def read_poses_bounds(path) -> Tuple[np.ndarray, np.ndarray, float]:
    """``poses_bounds.npy`` [N,17] -> poses [N,3,5] (3x4 pose | (H, W, focal) column), bounds [N,2] and the
    focal length of the first frame (src/data/LLFFDataset.py:24-27)."""
    table = np.load(path)
    poses = table[:, :15].reshape(-1, 3, 5)
    return poses, table[:, 15:], poses[0, 2, 4]


def list_frames(image_dir: Path) -> List[Path]:
    """Image files of a scene directory, per suffix in sorted order (src/data/LLFFDataset.py:30-36)."""
    if not image_dir.exists():
        raise FileNotFoundError(f"Image directory {image_dir} not found.")
    frames = [p for suffix in IMAGE_SUFFIXES for p in sorted(image_dir.glob("*." + suffix))]
    if not frames:
        raise FileNotFoundError(f"No images found in {image_dir}")
    return frames


def split_indices(total: int, split: str) -> np.ndarray:
    """The reference holds out nothing: 'train' is every frame, 'val' is empty (src/data/LLFFDataset.py:43-54)."""
    if split not in ("train", "val"):
        raise ValueError(f"Unknown split: {split}")
    return np.arange(total if split == "train" else 0)


def shrink_transposed(images: torch.Tensor, scale: float) -> torch.Tensor:
    """[N,H,W,3] -> [N,W',H',3] with H' = int(H scale), W' = int(W scale): bilinear with antialiasing (what
    torchvision does to a tensor image), and the reference's permute(2, 1, 0) that leaves the result
    width-major (src/data/LLFFDataset.py:66-75)."""
    n, h, w, _ = images.shape
    small = torch.nn.functional.interpolate(images.permute(0, 3, 1, 2), size=(int(h * scale), int(w * scale)),
                                            mode="bilinear", align_corners=False, antialias=True)
    return small.permute(0, 3, 2, 1).contiguous()


class LLFFDataset(Dataset):
    """Same constructor, fields and items as the reference class (src/data/LLFFDataset.py:9-86): an item is
    (image, pose [3,4], focal, H, W, bounds [2]).  The reference's quirks are kept: every frame is a training
    frame, shrunk images come back transposed, and the focal length is left unscaled (its rescale statement
    sits inside a comment, :75)."""

    near, far = 2.0, 6.0

    def __init__(self, scene_dir, split="train", factor=8):
        self.scene_dir, self.split, self.factor = Path(scene_dir), split, factor
        poses, bounds, self.focal = read_poses_bounds(self.scene_dir / "poses_bounds.npy")
        frames = list_frames(self.scene_dir / f"images_{factor}")
        pixels = torch.from_numpy(np.stack([load_array(f) / 255.0 for f in frames])).float()   # [N,H,W,3]
        self.indices = split_indices(len(pixels), split)
        self.poses, self.bounds, self.images = poses[self.indices], bounds, pixels[self.indices]
        self.H, self.W = (int(v) for v in self.images.shape[1:3])
        self.resize_factor = TARGET_WIDTH / self.W
        if self.resize_factor < 1.0:
            self.images = shrink_transposed(self.images, self.resize_factor)
            self.H, self.W = int(self.H * self.resize_factor), int(self.W * self.resize_factor)

    def __len__(self):
        return len(self.images)

    def __getitem__(self, idx):
        return (self.images[idx], torch.from_numpy(self.poses[idx, :3, :4]).float(), self.focal, self.H, self.W,
                self.bounds[idx])
