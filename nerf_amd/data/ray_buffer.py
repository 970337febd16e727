"""A GPU-resident ray source over all images of a scene.

The reference trains one image at a time: a DataLoader item, a Camera, all of its rays
(src/train_nerf_skow.py:132-146, src/train_nerf_llff.py:106-111).  Here the poses, intrinsics and
pixels of every image stay in device memory and a batch is any set of (image, pixel) pairs, turned into
rays and target colours by one kernel; batches are drawn from a shuffled order of all pixels of all
images, sharded across ranks without any exchange.
"""
from __future__ import annotations

from typing import Optional, Sequence

import torch

from .. import ops
from .._lib import require as _require


def camera_rows(poses: torch.Tensor, intrinsics: torch.Tensor) -> torch.Tensor:
    """poses [n,3,4] or [n,4,4], intrinsics [n,4] = (fx, fy, cx, cy) -> the [n,16] table of raygen_multi."""
    n = poses.shape[0]
    return torch.cat([poses[:, :3, :4].reshape(n, 12).float(), intrinsics.reshape(n, 4).float()], dim=1).contiguous()


class RayBuffer:
    def __init__(self, images: torch.Tensor, poses: torch.Tensor, intrinsics: torch.Tensor, device,
                 rank: int = 0, world: int = 1, seed: int = 0):
        """images [n,H,W,3] in [0,1]; poses [n,3,4] (or [n,4,4]); intrinsics [n,4] = (fx, fy, cx, cy)."""
        _require(images.dim() == 4 and images.shape[-1] == 3, "images must be [n,H,W,3]")
        self.n, self.H, self.W = images.shape[0], images.shape[1], images.shape[2]
        self.images = images.to(device=device, dtype=torch.float32).contiguous()
        self.cams = camera_rows(poses, intrinsics).to(device)
        self.device = torch.device(device)
        self.rank, self.world = rank, world
        self.gen = torch.Generator(device="cpu").manual_seed(seed)   # the same order on every rank
        self.total = self.n * self.H * self.W
        self._order: Optional[torch.Tensor] = None
        self._pos = 0

    @classmethod
    def from_capture(cls, dataset, device, **kw):
        """All frames of a NeRFCaptureDataset split, decoded at its ``img_wh``."""
        imgs, poses, intr = [], [], []
        for i in range(len(dataset)):
            img, pose, focal, pp, _ = dataset[i]
            imgs.append(img.permute(1, 2, 0))
            poses.append(pose[:3, :4])
            intr.append(torch.cat([focal, pp]))
        return cls(torch.stack(imgs), torch.stack(poses), torch.stack(intr), device, **kw)

    def rays_of(self, cam_id: torch.Tensor, pix: torch.Tensor):
        """rays_o, rays_d, rgb [R,3] of the given (image, pixel id = row*W + col) pairs."""
        return ops.raygen_multi(self.cams, self.H, self.W, cam_id.to(self.device), pix.to(self.device), self.images)

    def _reshuffle(self):
        # drawn on the host (the same order on every rank, whatever its GPU) and kept on the device: one copy per epoch;
        # a batch is then cut and split into (image, pixel) on the device, with no host-to-device copy per step
        self._order = torch.randperm(self.total, generator=self.gen).to(self.device)
        self._pos = 0

    def next_batch(self, rays: int):
        """The next ``rays`` rays of this rank from the shuffled order (a new order when it runs out).
        A global batch of world * rays consecutive entries is cut into one slice per rank."""
        span = rays * self.world
        if self._order is None or self._pos + span > self.total:
            self._reshuffle()
        flat = self._order[self._pos + self.rank * rays: self._pos + (self.rank + 1) * rays]
        self._pos += span
        hw = self.H * self.W
        cam_id = torch.div(flat, hw, rounding_mode="floor")
        pix = flat - cam_id * hw
        return self.rays_of(cam_id.to(torch.int32), pix.to(torch.int32))
