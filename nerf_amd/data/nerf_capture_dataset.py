"""NeRFCapture scenes: ``transforms.json`` + PNG frames (src/data/nerf_capture_dataset.py)."""
from __future__ import annotations

import json
import os
import random
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

from ._images import load_rgb_resized


def read_transforms(scene_dir: str) -> dict:
    """The scene's ``transforms.json`` (src/data/nerf_capture_dataset.py:20-21)."""
    with open(os.path.join(scene_dir, "transforms.json"), "r") as f:
        return json.load(f)


def normalize_poses(poses: Sequence[torch.Tensor], margin: float = 1.2) -> Tuple[torch.Tensor, torch.Tensor]:
    """Move the camera centres into the unit sphere, in place: subtract their mean and divide by
    ``margin`` times the largest distance from it (src/data/nerf_capture_dataset.py:52-62).
    Returns (scene_center [3], scene_scale [])."""
    centers = torch.stack([pose[:3, 3] for pose in poses])
    scene_center = centers.mean(0)
    dists = (centers - scene_center).norm(dim=-1)
    scene_scale = margin * dists.max()
    for pose in poses:
        pose[:3, 3] = (pose[:3, 3] - scene_center) / scene_scale
    return scene_center, scene_scale


class NeRFCaptureDataset(Dataset):
    """Same constructor, fields and items as the reference class (src/data/nerf_capture_dataset.py:11-111):
    an item is (img [3,H,W], pose [4,4], focal [2], principal_point [2], img_wh [2]) with the intrinsics
    rescaled from the frame's own size to ``img_wh``.  ``transform``, when given, receives a PIL image."""

    def __init__(self, scene_dir, split="train", img_wh=(100, 75), transform=None, subset=None):
        super().__init__()
        self.scene_dir = scene_dir
        self.img_wh = img_wh
        self.transform = transform
        self.meta = read_transforms(scene_dir)
        self.split = split
        frames = self.meta["frames"]
        if subset is not None:
            frames = random.sample(frames, subset)
        self.frames = frames
        self.image_paths: List[str] = [os.path.join(scene_dir, fr["file_path"] + ".png") for fr in frames]
        self.cx = [fr["cx"] for fr in frames]
        self.cy = [fr["cy"] for fr in frames]
        self.fl_x = [fr["fl_x"] for fr in frames]
        self.fl_y = [fr["fl_y"] for fr in frames]
        self.w = [fr["w"] for fr in frames]
        self.h = [fr["h"] for fr in frames]
        self.poses = [torch.from_numpy(np.array(fr["transform_matrix"])).float() for fr in frames]
        self.scene_center, self.scene_scale = normalize_poses(self.poses)
        self.N_images = len(frames)
        # 10 % of the frames, drawn at random, are the validation split (:73-79)
        perm = torch.randperm(len(self.poses))
        val_count = int(len(perm) * 0.10)
        self.val_idx = perm[:val_count].tolist()
        self.train_idx = perm[val_count:].tolist()

    def __len__(self):
        return len(self.train_idx) if self.split == "train" else len(self.val_idx)

    def scaled_intrinsics(self, real_idx: int) -> Tuple[float, float, float, float]:
        """(fx, fy, cx, cy) of frame ``real_idx`` at the dataset's ``img_wh`` (:94-103)."""
        new_w, new_h = self.img_wh
        scale_w = new_w / self.w[real_idx]
        scale_h = new_h / self.h[real_idx]
        return (self.fl_x[real_idx] * scale_w, self.fl_y[real_idx] * scale_h,
                self.cx[real_idx] * scale_w, self.cy[real_idx] * scale_h)

    def real_index(self, idx: int) -> int:
        return self.train_idx[idx] if self.split == "train" else self.val_idx[idx]

    def __getitem__(self, idx):
        real_idx = self.real_index(idx)
        if self.transform is None:
            img = load_rgb_resized(self.image_paths[real_idx], self.img_wh[0], self.img_wh[1])
        else:
            from PIL import Image
            img = self.transform(Image.open(self.image_paths[real_idx]).convert("RGB"))
        fx, fy, cx, cy = self.scaled_intrinsics(real_idx)
        new_w, new_h = self.img_wh
        focal = torch.tensor([fx, fy], dtype=torch.float32)
        principal_point = torch.tensor([cx, cy], dtype=torch.float32)
        img_wh = torch.tensor([new_w, new_h], dtype=torch.float32)
        return img, self.poses[real_idx], focal, principal_point, img_wh
