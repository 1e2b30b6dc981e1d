"""Python faces of the E-step entry points of include/beer_hip.h.

Shapes and argument checks only: the arithmetic is in beer_amd/csrc.  All
returned tensors live on the GPU.
"""

import os

import torch

from . import _hip
from .stats import FrameStats

__all__ = ['normal_llh', 'mixtureset_estep', 'normal_accumulate', 'weights_from_acc',
           'is_dense', 'dense_llh', 'dense_softmax', 'dense_accumulate', 'rowdot',
           'attach_stats_grad', 'differentiable_stats', 'sample_stats', 'attach_frame_grad',
           'frames_llh_backward', 'normal_llh_autograd']

LOG_2PI = 1.8378770664093453


def is_dense(stats):
    'True for dense [T, Q] statistics (the VAE "stats-in" path).'
    return isinstance(stats, torch.Tensor)


def _exact(X):
    '''Whether this call multiplies float32 frames on the exact fp32 MFMA: small
    inputs, or the caller asked for it (`_hip.f32_fast_ok`).'''
    return X.dtype == torch.float32 and not _hip.f32_fast_ok(X)


def _frames(stats):
    if not isinstance(stats, FrameStats):
        raise TypeError('expected the lazy statistics returned by '
                        'model.sufficient_statistics(X)')
    return stats


def normal_llh(stats, exp_stats, cov_type):
    'l[t,k] = scale * phi(x_t) . E[T]_k - D/2 ln 2pi -> [T, K].'
    st = _frames(stats)
    X = st.data
    T, D = X.shape
    E = _hip.on_device(exp_stats, X.dtype)
    K = E.shape[0]
    if st.scale == 1.0:
        # K mixtures of one component: the log-normalisers ARE the log-likelihoods, and that
        # form of the call runs on the matrix cores (the per-component output does not)
        return mixtureset_estep(st, E, None, K, 1, cov_type, want_resps=False)[0]
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
    ws, ws_bytes = _hip.workspace('beer_estep_workspace_bytes', X.dtype,
                                  _hip.COV_CODE[cov_type], D, S, G, X.device)
    # the generic kernels normalise in place in the responsibilities buffer; the
    # matrix-core kernels keep them in registers.  Those take group-aligned shapes
    # (one mixture, or G a power of two) in every arithmetic, and any G on the
    # float32 bf16x3 path when no responsibilities are wanted (padded groups).
    exact = _exact(X)
    aligned = S == 1 or (G & (G - 1)) == 0
    on_matrix_cores = ws is not None and labels is None and st.scale == 1.0 and \
        (aligned or (X.dtype == torch.float32 and not exact and not want_resps))
    need_resps = want_resps or (G > 1 and not on_matrix_cores)
    resps = torch.empty(T, K, dtype=X.dtype, device=X.device) if need_resps else None
    lab = None
    if labels is not None:
        lab = _hip.on_device(torch.as_tensor(labels)).to(torch.int64).contiguous()
    if not need_resps and on_matrix_cores and S > 1 and G >= 4 and not exact and \
            X.dtype == torch.float32 and _hip.f32_fast_ok(X):
        # log-normalisers only, on the bf16x3 path: the logits' A fragments from the frame
        # fragment image where the frames have one (the fused accumulation uses the same)
        img = st.frame_image(cov_type)
        if img is not None:
            try:
                _hip.call('beer_mixtureset_lognorm_image', _hip.COV_CODE[cov_type], T, D, S, G,
                          _hip.ptr(X), _hip.ptr(E), _hip.ptr(lw), _hip.ptr(img),
                          _hip.ptr(log_norm), _hip.ptr(llh_sum), _hip.ptr(ws), ws_bytes)
                return log_norm, None
            except _hip.HipInvalid:
                pass          # a shape the image kernels do not take (refused before any launch)

    def launch(resps):
        _hip.call('beer_mixtureset_estep', _hip.dtype_code(X.dtype, exact),
                  _hip.COV_CODE[cov_type], T, D, S, G, _hip.ptr(X), _hip.ptr(E), _hip.ptr(lw),
                  _hip.ptr(lab), st.scale, None, _hip.ptr(log_norm), _hip.ptr(resps),
                  _hip.ptr(llh_sum), _hip.ptr(ws), ws_bytes)
    try:
        launch(resps)
    except _hip.HipInvalid:
        # `on_matrix_cores` above restates the library's own dispatch; should the two
        # ever disagree, the library refuses a G > 1 call without a responsibilities
        # buffer (the generic kernels normalise in it): give it one
        if resps is not None or G == 1:
            raise
        resps = torch.empty(T, K, dtype=X.dtype, device=X.device)
        launch(resps)
        if not want_resps:
            resps = None
    return log_norm, resps


class PackedResps:
    '''Responsibilities [T, K] of one mixture in the form the bf16x3 accumulation
    kernel multiplies with (include/beer_hip.h: beer_mixture_estep_packed):
    `words` int32, the three-plane tiles of 64 frames x 128 components that kernel
    copies to LDS.  `unpack()` gives the float32 matrix (exactly).'''

    def __init__(self, words, nframes, ncomp):
        self.words, self.shape = words, (nframes, ncomp)

    def unpack(self):
        T, K = self.shape
        out = torch.empty(T, K, dtype=torch.float32, device=self.words.device)
        _hip.call('beer_unpack_resps', T, K, _hip.ptr(self.words), _hip.ptr(out))
        return out


def packed_path_ok(stats, K, cov_type):
    '''True when the E-step of a K-component mixture over `stats` can hand its
    responsibilities to the accumulation in packed form: float32 frames on the
    bf16x3 path, unscaled, shapes with a matrix-core kernel on both sides.'''
    st = _frames(stats)
    X = st.data
    if st.scale != 1.0 or not _hip.f32_fast_ok(X):
        return False
    code, D = _hip.COV_CODE[cov_type], X.shape[1]
    lib, dt = _hip.lib(), _hip.dtype_code(X.dtype)
    # (the packed kernels' own queries: they take D <= 128, the exact ones D <= 96)
    return lib.beer_estep_workspace_bytes(dt, code, D, 1, K) > 0 and \
        lib.beer_accumulate_packed_workspace_bytes(code, X.shape[0], D, K) > 0


def mixture_estep_packed(stats, exp_stats, log_weights, K, cov_type, llh_sum=None):
    '''(log_norm [T,1], PackedResps) of one mixture: `mixtureset_estep` with
    S = 1 whose responsibilities go straight to `normal_accumulate`.  Only where
    `packed_path_ok`.'''
    st = _frames(stats)
    X = st.data
    T, D = X.shape
    E = _hip.on_device(exp_stats, X.dtype)
    lw = _hip.on_device(log_weights, X.dtype)
    if E.shape[0] != K or lw.numel() != K:
        raise ValueError(f'{E.shape[0]} Gaussians, {lw.numel()} weights for {K} components')
    log_norm = torch.empty(T, 1, dtype=X.dtype, device=X.device)
    words = torch.empty(_hip.lib().beer_packed_resps_bytes(T, D, K) // 4, dtype=torch.int32,
                        device=X.device)
    ws, ws_bytes = _hip.workspace('beer_estep_workspace_bytes', X.dtype,
                                  _hip.COV_CODE[cov_type], D, 1, K, X.device)
    _hip.call('beer_mixture_estep_packed', _hip.COV_CODE[cov_type], T, D, K, _hip.ptr(X),
              _hip.ptr(E), _hip.ptr(lw), _hip.ptr(log_norm), _hip.ptr(words), _hip.ptr(llh_sum),
              _hip.ptr(ws), ws_bytes)
    return log_norm, PackedResps(words, T, K)


def packed_sets_ok(stats, S, G, cov_type):
    '''True when a mixture SET can hand the responsibilities within its states'
    mixtures to the accumulation as packed tiles, the state posteriors being
    multiplied in by the accumulation kernel (include/beer_hip.h:
    beer_mixtureset_estep_packed): float32 frames on the bf16x3 path, full
    covariance, G a power of two in 8..128.'''
    st = _frames(stats)
    X = st.data
    if st.scale != 1.0 or X.dtype != torch.float32 or not _hip.f32_fast_ok(X):
        return False
    return bool(_hip.lib().beer_mixtureset_packed_supported(_hip.COV_CODE[cov_type], X.shape[1],
                                                            S, G))


def mixtureset_estep_packed(stats, exp_stats, log_weights, S, G, cov_type, llh_sum=None):
    '''(log_norm [T,S], PackedResps [T, S*G]) of a mixture set: `mixtureset_estep`
    whose responsibilities go to `normal_accumulate(..., state_resps)` in packed
    form.  Only where `packed_sets_ok`.'''
    st = _frames(stats)
    X = st.data
    T, D = X.shape
    K = S * G
    E = _hip.on_device(exp_stats, X.dtype)
    lw = None if log_weights is None else _hip.on_device(log_weights, X.dtype)
    if E.shape[0] != K or (lw is not None and lw.numel() != K):
        raise ValueError(f'{E.shape[0]} Gaussians for {S} x {G} components')
    log_norm = torch.empty(T, S, dtype=X.dtype, device=X.device)
    words = torch.empty(_hip.lib().beer_packed_resps_bytes(T, D, K) // 4, dtype=torch.int32,
                        device=X.device)
    ws, ws_bytes = _hip.workspace('beer_estep_workspace_bytes', X.dtype,
                                  _hip.COV_CODE[cov_type], D, S, G, X.device)
    _hip.call('beer_mixtureset_estep_packed', _hip.COV_CODE[cov_type], T, D, S, G, _hip.ptr(X),
              _hip.ptr(E), _hip.ptr(lw), _hip.ptr(log_norm), _hip.ptr(words), _hip.ptr(llh_sum),
              _hip.ptr(ws), ws_bytes)
    return log_norm, PackedResps(words, T, K)


# ---- one mixture with more than 256 components ---------------------------------------------

class WideResps:
    '''Responsibilities [T, K] of ONE mixture with more than 256 components, in the
    factored form the matrix-core kernels of a mixture set work with: the
    components are cut into S2 blocks of G2; `block_lse` [T, S2] are the blocks'
    log-sum-exps, `block_resps` [T, S2] = exp(block_lse - log_norm) the blocks' shares,
    and within a block r = exp(l - block_lse) is recomputed from the frames
    (diagonal / isotropic) or held as packed tiles (full covariance).  The softmax
    over K components is the two-level softmax of these.  When K is not S2 * G2 the
    last block is filled up with phantom components of weight exp(-1e30) = 0
    (`exp_stats` / `log_weights` are the padded arrays, `K` the real count).
    `dense()` gives the [T, K] matrix.'''

    def __init__(self, stats, exp_stats, log_weights, split, cov_type, block_lse, block_resps,
                 packed, K=None):
        self.stats, self.exp_stats, self.log_weights = stats, exp_stats, log_weights
        self.split, self.cov_type = split, cov_type
        self.block_lse, self.block_resps, self.packed = block_lse, block_resps, packed
        self.K = split[0] * split[1] if K is None else K

    def dense(self):
        S2, G2 = self.split
        within = self.packed.unpack() if self.packed is not None else mixtureset_estep(
            self.stats, self.exp_stats, self.log_weights, S2, G2, self.cov_type)[1]
        return (within * self.block_resps.repeat_interleave(G2, dim=1))[:, :self.K]


def wide_mixture_split(stats, K, cov_type):
    '''(S2, G2), S2 * G2 >= K, when a mixture of K > 256 components over `stats` runs on
    the matrix-core kernels as S2 blocks of G2 components (two-level softmax); None
    otherwise (small inputs, exact mode: the generic kernels take it).  An exact
    factorisation is preferred; any other K gets its last block padded.'''
    st = _frames(stats)
    X = st.data
    if K <= 256 or st.scale != 1.0 or X.dtype != torch.float32 or not _hip.f32_fast_ok(X):
        return None
    ok = (lambda S2, G2: packed_sets_ok(st, S2, G2, cov_type)) if cov_type == 'full' else \
        (lambda S2, G2: fused_accumulate_ok(st, S2, G2, cov_type))
    first = -(-K // 256)
    for S2 in range(first, min(K // 8, first + 64) + 1):
        if K % S2 == 0 and ok(S2, K // S2):
            return S2, K // S2
    for G2 in ((128, 64) if cov_type == 'full' else (256, 128)):
        S2 = -(-K // G2)
        if ok(S2, G2):
            return S2, G2
    return None


def wide_mixture_estep(stats, exp_stats, log_weights, K, cov_type, split):
    '''(log_norm [T, 1], WideResps) of one mixture over `split` = wide_mixture_split(...):
    Mixture.expected_log_likelihood (beer/models/mixture.py:70-93) for K > 256.'''
    S2, G2 = split
    st = _frames(stats)
    E = _hip.on_device(exp_stats, st.data.dtype)
    lw = _hip.on_device(log_weights, st.data.dtype).reshape(-1)
    pad = S2 * G2 - K
    if pad:
        # phantom components: any finite parameters, log-weight -1e30 (exp underflows to an
        # exact 0: their responsibilities and statistics are zeros; not -inf, which the
        # multi-piece products would turn into 0 * inf)
        E = torch.cat([E, E[:1].expand(pad, -1)]).contiguous()
        lw = torch.cat([lw, torch.full((pad,), -1.0e30, dtype=lw.dtype, device=lw.device)])
    lw = lw.reshape(S2, G2)
    packed = None
    if cov_type == 'full':
        lse, packed = mixtureset_estep_packed(st, E, lw, S2, G2, cov_type)
    else:
        lse, _ = mixtureset_estep(st, E, lw, S2, G2, cov_type, want_resps=False)
    log_norm, share = dense_softmax(lse, None, 1, S2)
    return log_norm, WideResps(st, E, lw, split, cov_type, lse, share, packed, K=K)


def pack_resps(stats, comp_resps, state_resps, S, G):
    '''PackedResps of float32 responsibilities [T, S*G] (times `state_resps`
    [T, S] broadcast over each state's G components).'''
    st = _frames(stats)
    X = st.data
    T, D = X.shape
    K = S * G
    cr = _hip.on_device(comp_resps, X.dtype)
    sr = None if state_resps is None else _hip.on_device(state_resps, X.dtype)
    words = torch.empty(_hip.lib().beer_packed_resps_bytes(T, D, K) // 4, dtype=torch.int32,
                        device=X.device)
    _hip.call('beer_pack_resps', T, D, S, G, _hip.ptr(X), _hip.ptr(cr), _hip.ptr(sr),
              _hip.ptr(words))
    return PackedResps(words, T, K)


def _repack_pays(stats, K, cov_type):
    '''Accumulating float32 responsibilities: with more than 32 statistic tiles
    (full covariance, D >= 22) the float32 kernel re-reads them four times at a
    fifth of the MFMA rate; packing them first (one read, one write) and running
    the packed kernel is faster (K = 1920, D = 40: 11.8 -> 6 ms per 500 k frames).'''
    X = stats.data
    if cov_type != 'full' or stats.shape[1] <= 512 or K % 4 or stats.scale != 1.0 or \
            not _hip.f32_fast_ok(X):
        return False
    return _hip.lib().beer_accumulate_packed_workspace_bytes(
        _hip.COV_CODE[cov_type], X.shape[0], X.shape[1], K) > 0


def normal_accumulate(stats, comp_resps, state_resps, S, G, cov_type, acc=None):
    '''acc[k,:] += sum_t comp_resps[t,k] * state_resps[t, k // G] * phi(x_t),
    fp64 [S*G, Q].  `comp_resps` may be the `PackedResps` of
    `mixture_estep_packed` (no state responsibilities then).'''
    st = _frames(stats)
    if st.scale != 1.0:
        raise ValueError('scaled statistics cannot be accumulated')
    X = st.data
    T, D = X.shape
    K = S * G
    Q = st.shape[1]
    if acc is None:
        acc = torch.zeros(K, Q, dtype=torch.float64, device=X.device)
    if isinstance(comp_resps, WideResps):
        # one mixture, K > 256: the blocks' shares play the state posteriors' part
        wr = comp_resps
        S2, G2 = wr.split
        if state_resps is not None or wr.K != K or wr.stats.data.data_ptr() != X.data_ptr():
            raise ValueError('factored responsibilities: one mixture, the frames they came from')
        # (a padded last block: the phantom components' rows are zeros, dropped here)
        full = acc if S2 * G2 == K else torch.zeros(S2 * G2, Q, dtype=torch.float64, device=X.device)
        if wr.packed is not None:
            normal_accumulate(st, wr.packed, wr.block_resps, S2, G2, cov_type, acc=full)
        else:
            mixtureset_accumulate_fused(st, wr.exp_stats, wr.log_weights, wr.block_lse,
                                        wr.block_resps, S2, G2, cov_type, acc=full)
        if full is not acc:
            acc += full[:K]
        return acc
    if isinstance(comp_resps, PackedResps) and state_resps is not None:
        # mixture set: the state posteriors are multiplied in by the kernel
        if tuple(comp_resps.shape) != (T, K) or not packed_sets_ok(st, S, G, cov_type):
            raise ValueError('packed responsibilities of a mixture set: same frames and '
                             'components, a shape with `packed_sets_ok`')
        sr = _hip.on_device(state_resps, X.dtype)
        if tuple(sr.shape) != (T, S):
            raise ValueError(f'state responsibilities {tuple(sr.shape)}, expected {(T, S)}')
        code = _hip.COV_CODE[cov_type]
        ws, ws_bytes = _hip.packed_workspace(code, T, D, K, X.device, sets=(S, G))
        _hip.call('beer_mixtureset_accumulate_packed', code, T, D, S, G, _hip.ptr(X),
                  _hip.ptr(comp_resps.words), _hip.ptr(sr), _hip.ptr(acc), _hip.ptr(ws), ws_bytes)
        return acc
    if isinstance(comp_resps, PackedResps):
        if tuple(comp_resps.shape) != (T, K):
            raise ValueError('packed responsibilities: same frames and components')
        ws, ws_bytes = _hip.packed_workspace(_hip.COV_CODE[cov_type], T, D, K, X.device)
        _hip.call('beer_normal_accumulate_packed', _hip.COV_CODE[cov_type], T, D, K,
                  _hip.ptr(X), _hip.ptr(comp_resps.words), _hip.ptr(acc), _hip.ptr(ws), ws_bytes)
        return acc
    if comp_resps is not None and X.dtype == torch.float32 and _repack_pays(st, K, cov_type):
        return normal_accumulate(st, pack_resps(st, comp_resps, state_resps, S, G), None, S, G,
                                 cov_type, acc=acc)
    cr = None if comp_resps is None else _hip.on_device(comp_resps, X.dtype)
    sr = None if state_resps is None else _hip.on_device(state_resps, X.dtype)
    ws, ws_bytes = _hip.frames_workspace(X.dtype, _exact(X), _hip.COV_CODE[cov_type], T, D, S, G,
                                         X.device)
    _hip.call('beer_normal_accumulate', _hip.dtype_code(X.dtype, _exact(X)),
              _hip.COV_CODE[cov_type], T, D, S, G, _hip.ptr(X), _hip.ptr(cr), _hip.ptr(sr),
              _hip.ptr(acc), _hip.ptr(ws), ws_bytes)
    return acc


def fused_accumulate_ok(stats, S, G, cov_type):
    '''True when `mixtureset_accumulate_fused` takes the call: float32 frames on the
    bf16x3 path, unscaled, diagonal / isotropic Gaussians (few statistics per
    Gaussian) -- the E-step then needs to leave no responsibilities behind.'''
    st = _frames(stats)
    X = st.data
    if st.scale != 1.0 or not _hip.f32_fast_ok(X):
        return False
    # the E-step that leaves the log-normalisers must be the matrix-core one: the
    # recomputed logits then carry the same roundings as the normalisers (against the
    # generic kernels' logits a 1e-5 mismatch does not cancel)
    code = _hip.COV_CODE[cov_type]
    return _hip.lib().beer_accumulate_fused_workspace_bytes(code, X.shape[1], S, G) > 0 and \
        _hip.lib().beer_estep_workspace_bytes(_hip.F32, code, X.shape[1], S, G) > 0


def mixtureset_accumulate_fused(stats, exp_stats, log_weights, log_norm, state_resps, S, G,
                                cov_type, acc=None):
    '''acc[k,:] += sum_t exp(l[t,k] - log_norm[t, k // G]) * state_resps[t, k // G] *
    phi(x_t), the component logits l recomputed from the frames (no [T, K] matrix).'''
    st = _frames(stats)
    X = st.data
    T, D = X.shape
    K = S * G
    E = _hip.on_device(exp_stats, X.dtype)
    lw = None if log_weights is None else _hip.on_device(log_weights, X.dtype)
    ln = _hip.on_device(log_norm, X.dtype)
    sr = None if state_resps is None else _hip.on_device(state_resps, X.dtype)
    if tuple(ln.shape) != (T, S) or (sr is not None and tuple(sr.shape) != (T, S)):
        raise ValueError(f'log_norm / state_resps must be [{T}, {S}]')
    if acc is None:
        acc = torch.zeros(K, st.shape[1], dtype=torch.float64, device=X.device)
    code = _hip.COV_CODE[cov_type]
    nbytes = _hip.lib().beer_accumulate_fused_workspace_bytes(code, D, S, G)
    key = ('accf', code, D, S, G, X.device, torch.cuda.current_stream().cuda_stream)
    ws = _hip._workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _hip._workspaces[key] = torch.empty(nbytes, dtype=torch.uint8, device=X.device)
    img = st.frame_image(cov_type)
    _hip.call('beer_mixtureset_accumulate_fused', code, T, D, S, G, _hip.ptr(X), _hip.ptr(E),
              _hip.ptr(lw), _hip.ptr(ln), _hip.ptr(sr), _hip.ptr(img), _hip.ptr(acc), _hip.ptr(ws),
              nbytes)
    return acc


def weights_from_acc(acc, S, G):
    'Mixture-weight statistics [S, G] (fp64) from accumulated Gaussian stats.'
    out = torch.zeros(S, G, dtype=torch.float64, device=acc.device)
    _hip.call('beer_weights_from_acc', S, G, acc.shape[1], _hip.ptr(acc), _hip.ptr(out))
    return out


# ---- dense ("stats-in") statistics: the prior of a VAE ------------------------
# beer/models/vae.py:63-89 hands the prior sample-averaged statistics [T, Q]
# and differentiates its expected log-likelihood w.r.t. them.

def _dense(stats, dtype=None):
    return _hip.on_device(stats.detach(), dtype)


def dense_llh(stats, exp_stats, dim):
    'stats @ E[T]^T - dim/2 ln 2pi -> [T, K] (no autograd graph).'
    st = _dense(stats)
    E = _hip.on_device(exp_stats, st.dtype)
    T, Q = st.shape
    if E.shape[1] != Q:
        raise ValueError(f'statistics of dimension {Q} for parameters of dimension {E.shape[1]}')
    out = torch.empty(T, E.shape[0], dtype=st.dtype, device=st.device)
    _hip.call('beer_dense_llh', _hip.dtype_code(st.dtype), T, Q, E.shape[0], _hip.ptr(st),
              _hip.ptr(E), -.5 * dim * LOG_2PI, _hip.ptr(out))
    return out


def dense_softmax(pc_llh, log_weights, S, G, want_resps=True):
    '(log_norm [T,S], resps [T,S*G]) of pc_llh [T,S*G] + log_weights [S,G].'
    pc = _hip.on_device(pc_llh)
    T = pc.shape[0]
    lw = None if log_weights is None else _hip.on_device(log_weights, pc.dtype)
    log_norm = torch.empty(T, S, dtype=pc.dtype, device=pc.device)
    resps = torch.empty(T, S * G, dtype=pc.dtype, device=pc.device) if want_resps else None
    _hip.call('beer_softmax_groups', _hip.dtype_code(pc.dtype), T, S, G, _hip.ptr(pc),
              _hip.ptr(lw), _hip.ptr(log_norm), _hip.ptr(resps))
    return log_norm, resps


def rowdot(a, b):
    'sum_k a[t,k] b[t,k] -> [T].'
    a = _hip.on_device(a)
    b = _hip.on_device(b, a.dtype)
    out = torch.empty(a.shape[0], dtype=a.dtype, device=a.device)
    _hip.call('beer_rowdot', _hip.dtype_code(a.dtype), a.shape[0], a.shape[1], _hip.ptr(a),
              _hip.ptr(b), _hip.ptr(out))
    return out


def dense_accumulate(stats, comp_resps, state_resps, S, G, acc=None):
    'acc[k,:] += sum_t comp_resps[t,k] state_resps[t,k//G] stats[t,:], fp64 [S*G, Q].'
    st = _dense(stats)
    T, Q = st.shape
    K = S * G
    if acc is None:
        acc = torch.zeros(K, Q, dtype=torch.float64, device=st.device)
    if comp_resps is None:
        comp_resps, state_resps, G = state_resps, None, 1
    if comp_resps is None:
        comp_resps = torch.ones(T, K, dtype=st.dtype, device=st.device)
    cr = _hip.on_device(comp_resps.detach(), st.dtype)
    sr = None if state_resps is None else _hip.on_device(state_resps.detach(), st.dtype)
    _hip.call('beer_dense_accumulate', _hip.dtype_code(st.dtype), T, K, Q, G, _hip.ptr(cr),
              _hip.ptr(sr), _hip.ptr(st), _hip.ptr(acc))
    return acc


def _llh_backward(weights, grad, exp_stats):
    w = _hip.on_device(weights)
    E = _hip.on_device(exp_stats, w.dtype)
    g = None if grad is None else _hip.on_device(grad, w.dtype)
    out = torch.empty(w.shape[0], E.shape[1], dtype=w.dtype, device=w.device)
    _hip.call('beer_dense_llh_backward', _hip.dtype_code(w.dtype), w.shape[0], w.shape[1],
              E.shape[1], _hip.ptr(w), _hip.ptr(g), _hip.ptr(E), _hip.ptr(out))
    return out


class _StatsGrad(torch.autograd.Function):
    """value[t] = sum_k weights[t,k] llh[t,k] (+ terms without gradient), with
    d value[t] / d stats[t] = sum_k weights[t,k] E[T]_k: what the reference's
    autograd derives for `(pc_llhs * resps).sum(-1)` with detached resps
    (mixture.py:79,92; hmm.py:81-87)."""

    @staticmethod
    def forward(ctx, stats, value, weights, exp_stats):
        ctx.save_for_backward(weights, exp_stats)
        ctx.home = (stats.device, stats.dtype)
        return value.clone()

    @staticmethod
    def backward(ctx, grad):
        weights, exp_stats = ctx.saved_tensors
        out = _llh_backward(weights, grad.contiguous(), exp_stats)
        return out.to(device=ctx.home[0], dtype=ctx.home[1]), None, None, None


def attach_stats_grad(stats, value, weights, exp_stats):
    'Give `value` [T] its gradient w.r.t. dense `stats` (no-op without grad).'
    if not (torch.is_grad_enabled() and stats.requires_grad):
        return value
    return _StatsGrad.apply(stats, value, weights.detach(), exp_stats.detach())


class _DenseLlh(torch.autograd.Function):
    'Differentiable stats @ E[T]^T + base -> [T, K].'

    @staticmethod
    def forward(ctx, stats, exp_stats, dim):
        ctx.save_for_backward(exp_stats)
        ctx.home = (stats.device, stats.dtype)
        return dense_llh(stats, exp_stats, dim)

    @staticmethod
    def backward(ctx, grad):
        exp_stats, = ctx.saved_tensors
        out = _llh_backward(grad.contiguous(), None, exp_stats)
        return out.to(device=ctx.home[0], dtype=ctx.home[1]), None, None


def dense_llh_autograd(stats, exp_stats, dim):
    if torch.is_grad_enabled() and stats.requires_grad:
        return _DenseLlh.apply(stats, exp_stats.detach(), dim)
    return dense_llh(stats, exp_stats, dim)


class _SuffStats(torch.autograd.Function):
    '''Mean over `ns` consecutive rows of phi(X) as a dense [T, Q] tensor,
    differentiable w.r.t. the frames [T * ns, D].'''

    @staticmethod
    def forward(ctx, data, cov_type, ns):
        X = _hip.on_device(data.detach())
        ctx.save_for_backward(X)
        ctx.cov_type, ctx.ns = cov_type, ns
        ctx.home = (data.device, data.dtype)
        T, D = X.shape[0] // ns, X.shape[1]
        Q = FrameStats(X[:1], cov_type).shape[1]
        out = torch.empty(T, Q, dtype=X.dtype, device=X.device)
        _hip.call('beer_suffstats_mean', _hip.dtype_code(X.dtype), _hip.COV_CODE[cov_type],
                  T, ns, D, _hip.ptr(X), _hip.ptr(out))
        return out

    @staticmethod
    def backward(ctx, grad):
        X, = ctx.saved_tensors
        g = _hip.on_device(grad, X.dtype)
        out = torch.empty_like(X)
        _hip.call('beer_suffstats_backward', _hip.dtype_code(X.dtype),
                  _hip.COV_CODE[ctx.cov_type], X.shape[0] // ctx.ns, ctx.ns, X.shape[1],
                  _hip.ptr(X), _hip.ptr(g), _hip.ptr(out))
        return out.to(device=ctx.home[0], dtype=ctx.home[1]), None, None


def differentiable_stats(data, cov_type, nsamples=1):
    '''Dense statistics carrying the autograd graph of `data` [T * nsamples,
    D]: phi(X) for nsamples = 1, else the mean of phi over each run of
    `nsamples` rows -> [T, Q] (vae.py:73-74 without the [T*ns, Q] tensor).'''
    if data.dim() != 2 or data.shape[0] % nsamples:
        raise ValueError('expected [n_frames * nsamples, dim] samples')
    return _SuffStats.apply(data, cov_type, int(nsamples))


# ---- one sample per frame: the prior of a VAE on the frame kernels --------------
# With nsamples = 1 the statistics a VAE hands its prior (vae.py:73-74) are phi(z_t) of the
# samples themselves: the prior runs its frame kernels on them -- no [T, Q] tensor -- and the
# gradient w.r.t. the samples is one product (csrc/sample_grad.hip).

def sample_stats(data, cov_type):
    """Lazy statistics of frames that carry an autograd graph: a `FrameStats` of their
    values whose `source` is the differentiable tensor.  Models that find a `source` give
    their expected log-likelihood its gradient w.r.t. it (`attach_frame_grad`)."""
    if data.dim() != 2:
        raise ValueError('expected [n_frames, dim] samples')
    st = FrameStats(data.detach(), cov_type)
    if torch.is_grad_enabled() and data.requires_grad:
        st.source = data
    return st


def has_source(stats):
    return isinstance(stats, FrameStats) and stats.source is not None and \
        torch.is_grad_enabled() and stats.source.requires_grad


def frames_llh_backward(stats, weights, grad, exp_stats):
    """grad[t] * sum_k weights[t,k] * d l_k(x_t) / d x_t -> [T, D]
    (`beer_frames_llh_backward`; `grad` None = 1)."""
    st = _frames(stats)
    X = st.data
    T, D = X.shape
    w = _hip.on_device(weights, X.dtype)
    E = _hip.on_device(exp_stats, X.dtype)
    g = None if grad is None else _hip.on_device(grad, X.dtype)
    K = E.shape[0]
    if tuple(w.shape) != (T, K) or E.shape[1] != st.shape[1]:
        raise ValueError(f'weights {tuple(w.shape)} / parameters {tuple(E.shape)} for '
                         f'{T} frames of dimension {D}')
    out = torch.empty_like(X)
    code, cov = _hip.dtype_code(X.dtype), _hip.COV_CODE[st.cov_type]
    nbytes = _hip.lib().beer_frames_llh_backward_workspace_bytes(code, cov, T, D, K)
    if X.dtype == torch.float32 and _hip.get_f32_mode() == 'exact':
        # the matrix-core kernels of this call are bf16x3; without a workspace the library
        # runs its thread-per-output kernel (float64 accumulation): exact, and slow
        nbytes = 0
    ws = None
    if nbytes:
        key = ('sgrad', cov, D, K, X.device, torch.cuda.current_stream().cuda_stream)
        ws = _hip._workspaces.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = _hip._workspaces[key] = torch.empty(nbytes, dtype=torch.uint8, device=X.device)
    _hip.call('beer_frames_llh_backward', code, cov, T, D, K, _hip.ptr(X), _hip.ptr(w),
              _hip.ptr(g), _hip.ptr(E), _hip.ptr(out), _hip.ptr(ws), nbytes)
    if st.scale != 1.0:
        out *= st.scale
    return out


class _FrameGrad(torch.autograd.Function):
    """`_StatsGrad` for statistics that are phi(x_t) of differentiable frames: value[t] =
    sum_k weights[t,k] llh[t,k] (+ terms without gradient) differentiated w.r.t. the
    frames, the chain statistics -> frames included."""

    @staticmethod
    def forward(ctx, source, value, weights, exp_stats, stats):
        ctx.save_for_backward(weights, exp_stats)
        ctx.stats = stats
        ctx.home = (source.device, source.dtype)
        return value.clone()

    @staticmethod
    def backward(ctx, grad):
        weights, exp_stats = ctx.saved_tensors
        out = frames_llh_backward(ctx.stats, weights, grad.contiguous(), exp_stats)
        return out.to(device=ctx.home[0], dtype=ctx.home[1]), None, None, None, None


def attach_frame_grad(stats, value, weights, exp_stats):
    'Give `value` [T] its gradient w.r.t. the source of `stats` (no-op without one).'
    if not has_source(stats):
        return value
    return _FrameGrad.apply(stats.source, value, weights.detach(), exp_stats.detach(),
                            stats.detach())


class _FrameLlh(torch.autograd.Function):
    'Differentiable l[t,k] = phi(x_t) . E[T]_k + base -> [T, K] w.r.t. the frames.'

    @staticmethod
    def forward(ctx, source, exp_stats, stats):
        ctx.save_for_backward(exp_stats)
        ctx.stats = stats
        ctx.home = (source.device, source.dtype)
        return normal_llh(stats, exp_stats, stats.cov_type)

    @staticmethod
    def backward(ctx, grad):
        exp_stats, = ctx.saved_tensors
        out = frames_llh_backward(ctx.stats, grad.contiguous(), None, exp_stats)
        return out.to(device=ctx.home[0], dtype=ctx.home[1]), None, None


def normal_llh_autograd(stats, exp_stats, cov_type):
    '`normal_llh`, differentiable w.r.t. the source of `stats` when it has one.'
    if has_source(stats):
        return _FrameLlh.apply(stats.source, exp_stats.detach(), stats.detach())
    return normal_llh(stats, exp_stats, cov_type)
