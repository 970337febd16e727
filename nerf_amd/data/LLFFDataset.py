"""LLFF scenes: ``poses_bounds.npy`` + ``images_<factor>/`` (src/data/LLFFDataset.py)."""
from __future__ import annotations

from pathlib import Path
from typing import Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

from ._images import load_array


def read_poses_bounds(path) -> Tuple[np.ndarray, np.ndarray, float]:
    """``poses_bounds.npy`` [N,17] -> poses [N,3,5] (3x4 pose | (H, W, focal) column), bounds [N,2] and the
    focal length of the first frame (src/data/LLFFDataset.py:24-27)."""
    poses_bounds = np.load(path)
    poses = poses_bounds[:, :-2].reshape([-1, 3, 5])
    bounds = poses_bounds[:, -2:]
    return poses, bounds, poses[0, -1, -1]


class LLFFDataset(Dataset):
    """Same constructor, fields and items as the reference class (src/data/LLFFDataset.py:9-86): an item is
    (image, pose [3,4], focal, H, W, bounds [2]).  Like the reference, every frame is a training frame
    (its validation list is empty), images wider than 100 pixels are resized to width 100 and come back
    transposed ([W,H,3], the reference's permute(2, 1, 0)), and the focal length is left unscaled."""

    def __init__(self, scene_dir, split="train", factor=8):
        self.scene_dir = Path(scene_dir)
        self.split = split
        self.factor = factor
        self.near = 2.0
        self.far = 6.0
        self.poses, self.bounds, self.focal = read_poses_bounds(self.scene_dir / "poses_bounds.npy")
        image_dir = self.scene_dir / f"images_{factor}"
        if not image_dir.exists():
            raise FileNotFoundError(f"Image directory {image_dir} not found.")
        image_paths = (sorted(image_dir.glob("*.png")) + sorted(image_dir.glob("*.jpg")) +
                       sorted(image_dir.glob("*.PNG")) + sorted(image_dir.glob("*.JPG")))
        if not image_paths:
            raise FileNotFoundError(f"No images found in {image_dir}")
        images = np.stack([load_array(p) / 255.0 for p in image_paths], axis=0)   # [N,H,W,3]
        self.images = torch.from_numpy(images).float()
        total = len(self.images)
        if split == "train":
            self.indices = np.arange(total)
        elif split == "val":
            self.indices = np.arange(0)
        else:
            raise ValueError(f"Unknown split: {split}")
        self.poses = self.poses[self.indices]
        self.images = self.images[self.indices]
        self.H, self.W = self.images.shape[1:3]
        self.resize_factor = 100 / self.W
        if self.resize_factor < 1.0:
            new_H = int(self.H * self.resize_factor)
            new_W = int(self.W * self.resize_factor)
            # bilinear with antialiasing, as torchvision resizes a tensor image
            resized = torch.nn.functional.interpolate(self.images.permute(0, 3, 1, 2), size=(new_H, new_W),
                                                      mode="bilinear", align_corners=False, antialias=True)
            self.images = resized.permute(0, 3, 2, 1).contiguous()   # [N,W,H,3], see the class note
            self.H = int(self.H * self.resize_factor)
            self.W = int(self.W * self.resize_factor)

    def __len__(self):
        return len(self.images)

    def __getitem__(self, idx):
        pose = self.poses[idx, :3, :4]
        return self.images[idx], torch.from_numpy(pose).float(), self.focal, self.H, self.W, self.bounds[idx]
