"""Drop-in for src/utils/positional_encoding.py: sin/cos encoding on the GPU."""
import torch
import torch.nn as nn

from .. import ops


class PositionalEncoding(nn.Module):
    """gamma(x) = [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(L-1) x), cos(2^(L-1) x)].

    Constructor and attributes follow src/utils/positional_encoding.py:6-18
    (``freq_bands`` is a plain attribute, not a buffer, so it never appears in a
    state dict).  The HIP kernel covers the configuration every model of the
    reference uses (log-scale bands, input included); other settings raise.
    """

    def __init__(self, num_freqs=10, input_dims=3, include_input=True, log_scale=True):
        super().__init__()
        self.num_freqs = num_freqs
        self.include_input = include_input
        self.log_scale = log_scale
        if log_scale:
            self.freq_bands = 2.0 ** torch.linspace(0, num_freqs - 1, num_freqs)
        else:
            self.freq_bands = torch.linspace(2.0 ** 0, 2.0 ** (num_freqs - 1), num_freqs)
        self.output_dims = input_dims * (2 * num_freqs + (1 if include_input else 0))

    def forward(self, x):
        if not (self.log_scale and self.include_input):
            raise NotImplementedError("the HIP encoder implements log_scale=True, include_input=True only")
        self.freq_bands = self.freq_bands.to(x.device)
        return ops.encode_pe(x, self.num_freqs)
