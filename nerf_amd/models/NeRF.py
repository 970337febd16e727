"""Drop-in for src/models/NeRF.py: the 8x256 NeRF MLP as one fused HIP op."""
import torch
import torch.nn as nn

from .. import ops
from ..utils.positional_encoding import PositionalEncoding


class NeRF(nn.Module):
    """Same constructor, parameter names/shapes/creation order (hence the same
    seeded initial weights and state dict) as src/models/NeRF.py:16-65.

    forward(x [N,3], view_dirs [N,3]) -> rgb [N,3], sigma [N,1]; the layer-4
    skip input is [x_enc | h].  The HIP kernel implements the architecture
    every script of the reference instantiates (10/4 frequency bands, 8 layers
    of 256, skip at 4); other hyper-parameters raise NotImplementedError.
    """

    _skip_h_first = False  # layer-4 input is [x_enc | h]

    def __init__(self, input_dim=3, xyz_freqs=10, dir_freqs=4, hidden_dim=256, num_layers=8, skips=[4]):
        super().__init__()
        if (input_dim, xyz_freqs, dir_freqs, hidden_dim, num_layers, list(skips)) != (3, 10, 4, 256, 8, [4]):
            raise NotImplementedError("the fused kernel covers NeRF(3, 10, 4, 256, 8, [4]) only")
        self.xyz_freqs = xyz_freqs
        self.dir_freqs = dir_freqs
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.num_layers = num_layers
        self.skip_connections = skips
        self.pe_xyz = PositionalEncoding(num_freqs=xyz_freqs, input_dims=3)
        self.pe_dir = PositionalEncoding(num_freqs=dir_freqs, input_dims=3)
        self.input_xyz_dim = self.pe_xyz.output_dims
        self.input_dir_dim = self.pe_dir.output_dims
        self.layers_xyz = nn.ModuleList()
        for i in range(num_layers):
            if i == 0:
                width = self.input_xyz_dim
            elif i in skips:
                width = self.input_xyz_dim + hidden_dim
            else:
                width = hidden_dim
            self.layers_xyz.append(nn.Linear(width, hidden_dim))
        self.sigma_out = nn.Linear(hidden_dim, 1)
        self.feature_layer = nn.Linear(hidden_dim, hidden_dim)
        self.rgb_layers = nn.Sequential(nn.Linear(hidden_dim + self.input_dir_dim, hidden_dim // 2), nn.ReLU(),
                                        nn.Linear(hidden_dim // 2, 3), nn.Sigmoid())

    def _params(self):
        out = []
        for lin in self.layers_xyz:
            out += [lin.weight, lin.bias]
        for lin in (self.sigma_out, self.feature_layer, self.rgb_layers[0], self.rgb_layers[2]):
            out += [lin.weight, lin.bias]
        return out

    def forward(self, x, view_dirs):
        if view_dirs is None:
            # the reference returns zero colour and the density alone (NeRF.py:98-99)
            _, sigma = ops.vanilla_mlp(x, torch.zeros_like(x), self._params(), skip_h_first=False)
            return torch.zeros(x.shape[0], 3, device=x.device), sigma
        return ops.vanilla_mlp(x, view_dirs, self._params(), skip_h_first=False)
