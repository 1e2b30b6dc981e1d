"""Residual feed-forward encoder / decoder trunk of the VAE examples.

torch.nn modules (SURVEY.md marks beer/nnet as context of the hot path, not part of it;
the layers are `linear.Linear`: nn.Linear with a weight gradient that fills the chip).  Interface of
beer/nnet/residual.py:5-51: `ResidualFeedForwardNet(dim_in, nblocks, block_width)`
with `dim_in` / `dim_out`; the block class keeps the reference's spelling
(`ResidualFeedFowardBlock`) and sub-module names (`layer1`, `layer2`,
`activation_fn`, `blocks`) because pickled models and state dicts name them.
"""

import torch
from torch import nn

from .linear import Linear

__all__ = ['ResidualFeedForwardNet']


class ResidualFeedFowardBlock(nn.Module):
    'x -> x + act(W2 act(W1 x + b1) + b2): a bottleneck of `width` units around the identity.'

    def __init__(self, dim_in, width, activation_fn=nn.Tanh):
        super().__init__()
        self.layer1, self.layer2 = Linear(dim_in, width), Linear(width, dim_in)
        self.activation_fn = activation_fn()

    def forward(self, x):
        act = self.activation_fn
        return x + act(self.layer2(act(self.layer1(x))))


class ResidualFeedForwardNet(nn.Module):
    '`nblocks` residual blocks in sequence; the dimension does not change.'

    def __init__(self, dim_in, nblocks=1, block_width=10):
        super().__init__()
        self._dim_in = dim_in
        self.blocks = nn.Sequential()
        for n in range(nblocks):
            self.blocks.add_module(str(n), ResidualFeedFowardBlock(dim_in, block_width))

    dim_in = property(lambda self: self._dim_in)
    dim_out = property(lambda self: self._dim_in)

    def forward(self, X):
        return self.blocks(X)
