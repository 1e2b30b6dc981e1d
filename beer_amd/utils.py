"""Small helpers of the reference that the hot path uses
(beer/utils.py:84-123)."""

import torch

__all__ = ['onehot', 'logsumexp']


def onehot(labels, max_label, dtype, device):
    'One-hot encoding [len(labels), max_label] of a sequence of indices.'
    out = torch.zeros(len(labels), max_label, dtype=dtype, device=device)
    out[torch.arange(len(labels), device=device), torch.as_tensor(labels, device=device)] = 1
    return out


def logsumexp(tensor, dim=0):
    '-inf / +inf safe log-sum-exp along `dim`.'
    return torch.logsumexp(tensor, dim=dim)
