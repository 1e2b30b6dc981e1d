"""Python faces of the E-step entry points of include/beer_hip.h.

Shapes and argument checks only: the arithmetic is in beer_amd/csrc.  All
returned tensors live on the GPU.
"""

import torch

from . import _hip
from .stats import FrameStats

__all__ = ['normal_llh', 'mixtureset_estep', 'normal_accumulate', 'weights_from_acc']


def _frames(stats):
    if not isinstance(stats, FrameStats):
        raise NotImplementedError(
            'beer_amd E-step kernels take the lazy statistics returned by '
            'model.sufficient_statistics(X); dense [T, Q] statistics (VAE '
            '"stats-in" path) are a later row of the scope table')
    return stats


def normal_llh(stats, exp_stats, cov_type):
    'l[t,k] = scale * phi(x_t) . E[T]_k - D/2 ln 2pi -> [T, K].'
    st = _frames(stats)
    X = st.data
    T, D = X.shape
    E = _hip.on_device(exp_stats, X.dtype)
    K = E.shape[0]
    out = torch.empty(T, K, dtype=X.dtype, device=X.device)
    _hip.call('beer_mixtureset_estep', _hip.dtype_code(X.dtype), _hip.COV_CODE[cov_type],
              T, D, K, 1, _hip.ptr(X), _hip.ptr(E), None, None, st.scale,
              _hip.ptr(out), None, None, None, None, 0)
    return out


def mixtureset_estep(stats, exp_stats, log_weights, S, G, cov_type, labels=None,
                     want_resps=True, llh_sum=None):
    '''(log_norm [T,S], comp_resps [T,S*G] or None).  `log_weights` [S,G] or
    None.  `llh_sum`: optional fp64 device scalar, += sum log_norm.'''
    st = _frames(stats)
    X = st.data
    T, D = X.shape
    E = _hip.on_device(exp_stats, X.dtype)
    lw = None if log_weights is None else _hip.on_device(log_weights, X.dtype)
    K = S * G
    if E.shape[0] != K:
        raise ValueError(f'{E.shape[0]} Gaussians for {S} x {G} mixture components')
    log_norm = torch.empty(T, S, dtype=X.dtype, device=X.device)
    need_resps = want_resps or G > 1
    resps = torch.empty(T, K, dtype=X.dtype, device=X.device) if need_resps else None
    lab = None
    if labels is not None:
        lab = _hip.on_device(torch.as_tensor(labels)).to(torch.int64).contiguous()
    ws, ws_bytes = _hip.workspace('beer_estep_workspace_bytes', X.dtype,
                                  _hip.COV_CODE[cov_type], D, S, G, X.device)
    _hip.call('beer_mixtureset_estep', _hip.dtype_code(X.dtype), _hip.COV_CODE[cov_type],
              T, D, S, G, _hip.ptr(X), _hip.ptr(E), _hip.ptr(lw), _hip.ptr(lab), st.scale,
              None, _hip.ptr(log_norm), _hip.ptr(resps), _hip.ptr(llh_sum), _hip.ptr(ws),
              ws_bytes)
    return log_norm, resps


def normal_accumulate(stats, comp_resps, state_resps, S, G, cov_type, acc=None):
    '''acc[k,:] += sum_t comp_resps[t,k] * state_resps[t, k // G] * phi(x_t),
    fp64 [S*G, Q].'''
    st = _frames(stats)
    if st.scale != 1.0:
        raise ValueError('scaled statistics cannot be accumulated')
    X = st.data
    T, D = X.shape
    K = S * G
    Q = st.shape[1]
    if acc is None:
        acc = torch.zeros(K, Q, dtype=torch.float64, device=X.device)
    cr = None if comp_resps is None else _hip.on_device(comp_resps, X.dtype)
    sr = None if state_resps is None else _hip.on_device(state_resps, X.dtype)
    ws, ws_bytes = _hip.workspace('beer_accumulate_workspace_bytes', X.dtype,
                                  _hip.COV_CODE[cov_type], D, S, G, X.device)
    _hip.call('beer_normal_accumulate', _hip.dtype_code(X.dtype), _hip.COV_CODE[cov_type],
              T, D, S, G, _hip.ptr(X), _hip.ptr(cr), _hip.ptr(sr), _hip.ptr(acc),
              _hip.ptr(ws), ws_bytes)
    return acc


def weights_from_acc(acc, S, G):
    'Mixture-weight statistics [S, G] (fp64) from accumulated Gaussian stats.'
    out = torch.zeros(S, G, dtype=torch.float64, device=acc.device)
    _hip.call('beer_weights_from_acc', S, G, acc.shape[1], _hip.ptr(acc), _hip.ptr(out))
    return out
