"""Drop-in for src/utils/hash_grid_encoder.py: multiresolution hash grid on the GPU."""
import numpy as np
import torch
import torch.nn as nn

from .. import ops


class HashGridEncoder(nn.Module):
    """16-level hash grid with 2 features per level.

    Construction follows src/utils/hash_grid_encoder.py:18-37 so that a seeded
    build yields the reference's tables and the state dict has the same keys
    (``embeddings.<l>.weight`` and the int64 buffer ``primes``): geometric
    resolution schedule evaluated in float64 and truncated, min(2^T, res^3)
    rows per level, nn.Embedding default init followed by uniform(1e-4, 1e-3).
    """

    def __init__(self, n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                 base_resolution=16, finest_resolution=512):
        super().__init__()
        if n_features_per_level != 2:
            raise NotImplementedError("the HIP hash grid stores 2 features per row")
        self.n_levels = n_levels
        self.n_features_per_level = n_features_per_level
        growth = np.exp((np.log(finest_resolution) - np.log(base_resolution)) / (n_levels - 1))
        self.resolutions = [int(base_resolution * (growth ** i)) for i in range(n_levels)]
        self.hash_size = 2 ** log2_hashmap_size
        self.embeddings = nn.ModuleList()
        for res in self.resolutions:
            emb = nn.Embedding(min(self.hash_size, res ** 3), n_features_per_level)
            nn.init.uniform_(emb.weight, a=1e-4, b=1e-3)
            self.embeddings.append(emb)
        self.register_buffer('primes', torch.tensor([1, 2654435761, 805459861], dtype=torch.long))

    def hash_fn(self, coords, size):
        """Row of integer cell coordinates [...,3] (src/utils/hash_grid_encoder.py:39-40);
        host-visible helper, the kernels compute the same value in 32-bit arithmetic."""
        return ((coords * self.primes).sum(dim=-1) & 0xFFFFFFFF) % size

    def forward(self, x):
        """x [N,3] -> [N, n_levels*2]; differentiable in the tables."""
        return ops.hashgrid_encode(x, [e.weight for e in self.embeddings], self.resolutions)
