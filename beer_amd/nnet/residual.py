"""Residual feed-forward network (API of beer/nnet/residual.py:5-51)."""

import torch

__all__ = ['ResidualFeedForwardNet']


class ResidualFeedFowardBlock(torch.nn.Module):
    'y = x + f(W2 f(W1 x + b1) + b2).'

    def __init__(self, dim_in, width, activation_fn=torch.nn.Tanh):
        super().__init__()
        self.layer1 = torch.nn.Linear(dim_in, width)
        self.layer2 = torch.nn.Linear(width, dim_in)
        self.activation_fn = activation_fn()

    def forward(self, x):
        hidden = self.activation_fn(self.layer1(x))
        return x + self.activation_fn(self.layer2(hidden))


class ResidualFeedForwardNet(torch.nn.Module):
    'Stack of residual blocks; input and output have the same dimension.'

    def __init__(self, dim_in, nblocks=1, block_width=10):
        super().__init__()
        self._dim_in = dim_in
        self.blocks = torch.nn.Sequential(*[ResidualFeedFowardBlock(dim_in, block_width)
                                            for _ in range(nblocks)])

    @property
    def dim_in(self):
        return self._dim_in

    @property
    def dim_out(self):
        return self._dim_in

    def forward(self, X):
        return self.blocks(X)
