"""Offline ray precomputation for an LLFF scene (src/data/preload_llff_data.py): read ``poses_bounds.npy`` and
``images/``, split the frames 80 / 10 / 10 at random, and store per split the rays of every pixel
(``utils.get_rays.get_rays``: the older convention, +z forward and ``[W, H, 3]``), the colours and the per-pixel
near / far bounds as ``<split>.pt``.

Same functions, arguments and file contents as the reference's script; the rays come from the HIP ray generator
(``nerf_raygen`` behind ``utils.get_rays.get_rays``), so the pose must be moved to a GPU (``device``).  Images are
decoded with Pillow (imageio is not required).
"""
from __future__ import annotations

import os
from typing import Tuple

import numpy as np
import torch

from ..utils.get_rays import get_rays
from ._images import load_array


def load_llff_data(data_dir: str):
    """poses [N,3,5], bounds [N,2], images (list of float [H,W,C] tensors in [0,1]) -- src/data/preload_llff_data.py:15-25."""
    poses_bounds = np.load(os.path.join(data_dir, "poses_bounds.npy"))
    poses = poses_bounds[:, :15].reshape(-1, 3, 5)
    bounds = poses_bounds[:, 15:]
    img_dir = os.path.join(data_dir, "images")
    img_files = sorted(f for f in os.listdir(img_dir) if f.endswith(("png", "jpg")))
    images = [torch.from_numpy(load_array(os.path.join(img_dir, f)).astype(np.float32) / 255.0) for f in img_files]
    return poses, bounds, images


def split_dataset_indices(N: int, train_ratio=0.8, val_ratio=0.10) -> Tuple[list, list, list]:
    """A shuffle of range(N) by numpy's global generator, cut at 80 % and 90 % (src/data/preload_llff_data.py:28-33)."""
    indices = list(range(N))
    np.random.shuffle(indices)
    train_end = int(train_ratio * N)
    val_end = int((train_ratio + val_ratio) * N)
    return indices[:train_end], indices[train_end:val_end], indices[val_end:]


def precompute_and_split_llff(data_dir: str, out_dir: str, H=378, W=504, device="cuda"):
    """Write ``train.pt`` / ``val.pt`` / ``test.pt`` into ``out_dir`` (src/data/preload_llff_data.py:36-69): each a dict
    of ``rays_o``, ``rays_d`` [n,W,H,3], ``rgb`` (the images as loaded), ``near``, ``far`` [n,H,W] and ``focal``
    (the first frame's).  Tensors are stored on the CPU, as the reference stores them."""
    os.makedirs(out_dir, exist_ok=True)
    poses, bounds, images = load_llff_data(data_dir)
    train_idx, val_idx, test_idx = split_dataset_indices(len(images))
    for split, indices in {"train": train_idx, "val": val_idx, "test": test_idx}.items():
        rays_o_all, rays_d_all, rgbs_all, nears_all, fars_all = [], [], [], [], []
        for i in indices:
            c2w = torch.eye(4)
            # The reference assigns the whole [3,5] block here (src/data/preload_llff_data.py:49), which raises (the fifth
            # column -- H, W, focal -- does not fit a [3,4] slice): its evident intent, the 3x4 pose, is what is taken.
            c2w[:3, :4] = torch.from_numpy(poses[i][:, :4])
            focal = poses[i, -1, -1]
            rays_o, rays_d = get_rays(H, W, focal, c2w.to(device))
            rays_o_all.append(rays_o.cpu())
            rays_d_all.append(rays_d.cpu())
            rgbs_all.append(images[i])
            nears_all.append(torch.full((H, W), bounds[i, 0]))
            fars_all.append(torch.full((H, W), bounds[i, 1]))
        if not indices:                                   # torch.stack of an empty list raises in the reference too
            raise RuntimeError(f"split '{split}' is empty: the scene has too few frames")
        torch.save({"rays_o": torch.stack(rays_o_all), "rays_d": torch.stack(rays_d_all), "rgb": torch.stack(rgbs_all),
                    "near": torch.stack(nears_all), "far": torch.stack(fars_all), "focal": poses[0, -1, -1]},
                   os.path.join(out_dir, f"{split}.pt"))
