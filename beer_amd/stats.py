"""Lazy sufficient statistics.

The reference materialises phi(X) as a [T, Q] tensor on every call
(Q = D^2+D+2 = 1642 for full covariance at D = 40: 6.6 KB per frame) and
spends 68 % of a GMM step building it.  Here
`Model.sufficient_statistics(X)` returns this handle on the frames instead;
the E-step and accumulation kernels read X directly.  `.dense()` yields the
reference's tensor for callers that really want it.
"""

import torch

from . import _hip

__all__ = ['FrameStats', 'reference_layout', 'reference_layout_enabled']

_REFERENCE_LAYOUT = [False]


def reference_layout_enabled():
    return _REFERENCE_LAYOUT[0]


class reference_layout:
    '''Switch (also a context manager) to the reference's own return shapes where
    beer_amd normally keeps something cheaper:

    * `model.sufficient_statistics(X)` returns the dense `[T, Q]` tensor
      (beer/dists/normalwishart.py:30-38) instead of the lazy `FrameStats`;
    * `CompiledGraph.posteriors(llhs, trans_posteriors=True)` and
      `HMM.cache['trans_resps']` hold the transition posteriors per frame,
      `[T-1, S, S]` (beer/graph.py:308-323), instead of their sum over time.

    Meant for small inputs (notebooks, code that inspects these tensors): the
    statistics are 6.6 KB per frame at D = 40, the transition posteriors 8 S^2 bytes
    per frame.  `beer_amd.reference_layout(True)` switches it on for the process,
    `with beer_amd.reference_layout():` for a block.'''

    def __init__(self, enabled=True):
        self._previous = _REFERENCE_LAYOUT[0]
        _REFERENCE_LAYOUT[0] = bool(enabled)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        _REFERENCE_LAYOUT[0] = self._previous
        return False


class FrameStats:
    '''phi(X) * scale for a [T, D] block of frames, never formed unless asked.

    Behaves like the tensor the reference returns as far as the hot path uses
    it: `len()`, `.dtype`, `.device`, `.shape`, `stats * scale`.'''

    def __init__(self, data, cov_type, scale=1.0):
        if data.dim() != 2:
            raise ValueError('expected a [n_frames, dim] matrix of features')
        self.data = _hip.on_device(data)
        self.cov_type = cov_type
        self.scale = float(scale)

    def __len__(self):
        return self.data.shape[0]

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def device(self):
        return self.data.device

    @property
    def dim(self):
        return self.data.shape[1]

    @property
    def shape(self):
        D = self.data.shape[1]
        Q = {'full': D * D + D + 2, 'diagonal': 2 * D + 2, 'isotropic': D + 3}[self.cov_type]
        return torch.Size((self.data.shape[0], Q))

    def __mul__(self, scale):
        return FrameStats(self.data, self.cov_type, self.scale * float(scale))

    __rmul__ = __mul__

    def dense(self):
        'The [T, Q] tensor of the reference (beer_suffstats_expand).'
        T, D = self.data.shape
        out = torch.empty(tuple(self.shape), dtype=self.dtype, device=self.device)
        _hip.call('beer_suffstats_expand', _hip.dtype_code(self.dtype),
                  _hip.COV_CODE[self.cov_type], T, D, _hip.ptr(self.data), _hip.ptr(out))
        if self.scale != 1.0:
            out *= self.scale
        return out

    def __repr__(self):
        return (f'FrameStats(frames={len(self)}, dim={self.dim}, '
                f'cov_type={self.cov_type!r}, scale={self.scale})')
