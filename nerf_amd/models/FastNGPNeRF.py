"""Drop-in for src/models/FastNGPNeRF.py: hash-grid NeRF as one fused HIP op."""
import torch
import torch.nn as nn

from .. import ops
from ..utils.hash_grid_encoder import HashGridEncoder
from ..utils.positional_encoding import PositionalEncoding


class FastNGPNeRF(nn.Module):
    """Instant-NGP style model with the reference's quirks kept:

    * additive-prime hash on every level (dense ones too), floor-modulo wrap of
      negative coordinates (src/utils/hash_grid_encoder.py:39-40, 60-69);
    * sigma = softplus(raw[:, 0] + 1.5) on top of a density bias that is itself
      initialised to 1.5 (src/models/FastNGPNeRF.py:63, 91-92); sigma is [N];
    * view directions are re-normalised inside the model (:98);
    * colour hidden layer = fc_color[0]([feat | pe(dir)]) + skip_proj(feat) (:105-106).

    Parameters are created in the reference's order with its initialisers, so a
    seeded construction gives the same weights and the state dict has the same
    keys (incl. the int64 buffer ``hash_encoder.primes``).  The fused kernel is
    specialised to the default hyper-parameters' shapes (16 levels x 2 features,
    hidden width 64); table size and resolutions are free.
    """

    def __init__(self, n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16,
                 finest_resolution=512, hidden_dim=64):
        super().__init__()
        if (n_levels, n_features_per_level, hidden_dim) != (16, 2, 64):
            raise NotImplementedError("the fused kernel covers 16 levels x 2 features and hidden_dim=64 only")
        self.hash_encoder = HashGridEncoder(n_levels=n_levels, n_features_per_level=n_features_per_level,
                                            log2_hashmap_size=log2_hashmap_size, base_resolution=base_resolution,
                                            finest_resolution=finest_resolution)
        self.dir_encoder = PositionalEncoding(num_freqs=4, include_input=True)
        self.fc_density = nn.Sequential(nn.Linear(n_levels * n_features_per_level, hidden_dim), nn.ReLU(),
                                        nn.Linear(hidden_dim, 1 + 15))
        for layer in self.fc_density:
            if isinstance(layer, nn.Linear):
                nn.init.normal_(layer.weight, mean=0.0, std=1e-4)
                nn.init.constant_(layer.bias, 0)
        nn.init.constant_(self.fc_density[2].bias, 1.5)
        self.fc_color = nn.Sequential(nn.Linear(15 + self.dir_encoder.output_dims, hidden_dim), nn.ReLU(),
                                      nn.Linear(hidden_dim, 3))
        for layer in self.fc_color:
            if isinstance(layer, nn.Linear):
                nn.init.normal_(layer.weight, mean=0.0, std=1e-4)
                nn.init.constant_(layer.bias, 0)
        self.skip_proj = nn.Linear(15, hidden_dim, bias=False)

    def _tables(self):
        return [e.weight for e in self.hash_encoder.embeddings]

    def _params(self):
        return [self.fc_density[0].weight, self.fc_density[0].bias, self.fc_density[2].weight,
                self.fc_density[2].bias, self.fc_color[0].weight, self.fc_color[0].bias, self.fc_color[2].weight,
                self.fc_color[2].bias, self.skip_proj.weight]

    def forward(self, x, view_dirs=None, dirs_div=1, count=None):
        """x [N,3] (unit-cube / unit-sphere coordinates, may be negative), view_dirs [N,3] or None
        -> rgb [N,3], sigma [N].  ``dirs_div`` > 1 lets one direction row serve that many consecutive points.
        ``count`` (one int32 on the GPU): only the first ``count`` rows are in use (``ops.ngp_mlp``)."""
        return ops.ngp_mlp(x, view_dirs, self._tables(), self._params(), self.hash_encoder.resolutions, dirs_div, count)

    @torch.no_grad()
    def query_density(self, x):
        """sigma(x) without gradients (src/models/FastNGPNeRF.py:112-124)."""
        return self.forward(x, None)[1]
