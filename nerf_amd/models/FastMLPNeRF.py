"""Drop-in for src/models/FastMLPNeRF.py: same fused kernel as NeRF, with the
layer-4 skip input ordered [h | x_enc]."""
import torch.nn as nn

from .. import ops
from ..utils.positional_encoding import PositionalEncoding


class FastMLPNeRF(nn.Module):
    """Parameter names/shapes/creation order of src/models/FastMLPNeRF.py:7-45."""

    _skip_h_first = True  # layer-4 input is [h | x_enc]

    def __init__(self, num_layers=8, hidden_dim=256, skips=[4], num_encoding_fn_xyz=10, num_encoding_fn_dir=4):
        super().__init__()
        if (num_layers, hidden_dim, list(skips), num_encoding_fn_xyz, num_encoding_fn_dir) != (8, 256, [4], 10, 4):
            raise NotImplementedError("the fused kernel covers FastMLPNeRF(8, 256, [4], 10, 4) only")
        in_xyz = 3 * (2 * num_encoding_fn_xyz + 1)
        in_dir = 3 * (2 * num_encoding_fn_dir + 1)
        self.pe_xyz = PositionalEncoding(num_freqs=num_encoding_fn_xyz, input_dims=3)
        self.pe_dir = PositionalEncoding(num_freqs=num_encoding_fn_dir, input_dims=3)
        self.mlp_xyz = nn.ModuleList()
        for i in range(num_layers):
            if i == 0:
                width = in_xyz
            elif i in skips:
                width = hidden_dim + in_xyz
            else:
                width = hidden_dim
            self.mlp_xyz.append(nn.Linear(width, hidden_dim))
        self.fc_sigma = nn.Linear(hidden_dim, 1)
        self.fc_feature = nn.Linear(hidden_dim, hidden_dim)
        self.mlp_rgb = nn.Sequential(nn.Linear(hidden_dim + in_dir, hidden_dim // 2), nn.ReLU(),
                                     nn.Linear(hidden_dim // 2, 3), nn.Sigmoid())

    def _params(self):
        out = []
        for lin in self.mlp_xyz:
            out += [lin.weight, lin.bias]
        for lin in (self.fc_sigma, self.fc_feature, self.mlp_rgb[0], self.mlp_rgb[2]):
            out += [lin.weight, lin.bias]
        return out

    def forward(self, x, view_dirs):
        return ops.vanilla_mlp(x, view_dirs, self._params(), skip_h_first=True)
