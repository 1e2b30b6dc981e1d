"""`torch.nn.Linear` whose weight gradient over very many rows is a batched product.

The networks of a VAE are context of the hot path, plain torch.  One thing about
them is not plain on this hardware: the weight gradient `dW = dYᵀ X` of a layer applied to a
minibatch of a million frames is a [out, T] x [T, in] product with out, in <= 128 -- eight output
tiles for hipBLASLt, which then runs it on eight of 256 compute units (1.5-1.8 ms per layer at
T = 1 M, 24 of the 45 ms of a config-4 step).  Split over blocks of rows it is a batched product
that fills the chip (0.2-0.3 ms), and the shorter float32 sums are closer to the float64 result.
Forward pass, input gradient, parameters, state-dict names and pickles are nn.Linear's.
"""

import torch
from torch import nn
from torch.nn import functional as F

__all__ = ['Linear', 'weight_grad']

ROW_SPLIT_MIN = 1 << 16       # rows from which the weight gradient is split
ROW_BLOCK = 4096              # rows per block of the batched product


def weight_grad(grad_out, x, rows=None):
    '`grad_outᵀ @ x` ([T, out], [T, in] -> [out, in]) as a sum over blocks of `rows` rows.'
    rows = rows or ROW_BLOCK
    T = x.shape[0]
    nblocks = T // rows
    if nblocks < 2:
        return grad_out.t().mm(x)
    main = nblocks * rows
    gy, xx = grad_out.contiguous(), x.contiguous()
    g = torch.bmm(gy[:main].view(nblocks, rows, -1).transpose(1, 2),
                  xx[:main].view(nblocks, rows, -1)).sum(0)
    if main < T:
        g += gy[main:].t().mm(xx[main:])
    return g


class _RowSplitLinear(torch.autograd.Function):

    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return F.linear(x, weight, bias)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, grad_out):
        x, weight = ctx.saved_tensors
        need = ctx.needs_input_grad
        return (grad_out.mm(weight) if need[0] else None,
                weight_grad(grad_out, x) if need[1] else None,
                grad_out.sum(0) if ctx.has_bias and need[2] else None)


class Linear(nn.Linear):
    'nn.Linear; on a [T, in] input of at least `ROW_SPLIT_MIN` rows the weight gradient is row-split.'

    def forward(self, x):
        if x.dim() == 2 and x.shape[0] >= ROW_SPLIT_MIN and torch.is_grad_enabled() and \
                (self.weight.requires_grad or x.requires_grad):
            return _RowSplitLinear.apply(x, self.weight, self.bias)
        return F.linear(x, self.weight, self.bias)
