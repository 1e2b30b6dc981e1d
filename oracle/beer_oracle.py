"""CPU oracle for the beer variational-Bayes hot path.

TEST INFRASTRUCTURE ONLY.  This module is a numpy restatement of the reference
algorithm (beer-asr/beer, pure Python/torch).  It may be imported by `tests/`,
by `__graft_entry__.smoke()` and by the `cpu_baseline` leg of `bench.py` and by
nothing else: the product (`beer_amd`) never routes through it.

Parity status: PINNED.  The reference's own tests hold no number for this path
(they are stale), so the oracle is pinned against
outputs of the reference itself: `tests/golden/make_golden.py` imports
`/root/reference` beer in the build container and commits inputs+outputs as
`tests/golden/*.npz`; `tests/test_oracle_golden.py` checks every function here
against them (fp64 <= 1e-10 relative).

Every function cites the reference file:line it restates (paths relative to
the reference root).  Arrays keep the dtype they come in with (float32 or
float64), as the reference does.
"""

import math

import numpy as np
from scipy.special import digamma, gammaln, logsumexp as _sp_logsumexp

LOG2PI = math.log(2 * math.pi)


# ---------------------------------------------------------------------------
# Likelihood sufficient statistics  (K1)
# ---------------------------------------------------------------------------

def suffstats_full(X):
    """beer/dists/normalwishart.py:30-38 -- [x, -.5 vec(xx^T), -.5, .5]."""
    T = len(X)
    quad = X[:, :, None] * X[:, None, :]
    one = np.ones((T, 1), dtype=X.dtype)
    return np.concatenate([X, -.5 * quad.reshape(T, -1), -.5 * one, .5 * one],
                          axis=-1)


def suffstats_diag(X):
    """beer/dists/normalgamma.py:20-27 -- [x, -.5 x^2, -.5, .5]."""
    one = np.ones((len(X), 1), dtype=X.dtype)
    return np.concatenate([X, -.5 * X ** 2, -.5 * one, .5 * one], axis=-1)


def suffstats_iso(X):
    """beer/dists/isonormalgamma.py:21-30 -- [x, -.5 |x|^2, -.5, .5 D]."""
    D = X.shape[-1]
    one = np.ones((len(X), 1), dtype=X.dtype)
    return np.concatenate([X, -.5 * np.sum(X ** 2, axis=-1).reshape(-1, 1),
                           -.5 * one, .5 * D * one], axis=-1)


SUFFSTATS = {'full': suffstats_full, 'diagonal': suffstats_diag,
             'isotropic': suffstats_iso}


def stats_dim(cov_type, D):
    return {'full': D * D + D + 2, 'diagonal': 2 * D + 2,
            'isotropic': D + 3}[cov_type]


# ---------------------------------------------------------------------------
# Normal-Wishart (full covariance)
# ---------------------------------------------------------------------------

def _chol_logdet(W):
    L = np.linalg.cholesky(W)
    return 2 * np.log(np.diagonal(L, axis1=-2, axis2=-1)).sum(-1, keepdims=True)


def nw_expected_stats(mean, scale, scale_matrix, dof):
    """beer/dists/normalwishart.py:170-210.  mean [K,D], scale [K,1],
    scale_matrix [K,D,D], dof [K,1] -> E_q[T] [K, D^2+D+2]."""
    K, D = mean.shape
    idxs = np.arange(1, D + 1, dtype=mean.dtype)
    logdet = _chol_logdet(scale_matrix)
    mean_quad = mean[:, :, None] * mean[:, None, :]
    exp_prec = dof[:, :, None] * scale_matrix
    tr = (exp_prec.reshape(K, -1) * mean_quad.reshape(K, -1)).sum(-1, keepdims=True)
    return np.concatenate([
        np.matmul(exp_prec, mean[:, :, None]).reshape(K, D),
        exp_prec.reshape(K, D * D),
        (D / scale) + tr,
        digamma(.5 * (dof + 1 - idxs)).sum(-1, keepdims=True)
        + D * math.log(2) + logdet,
    ], axis=-1).astype(mean.dtype)


def nw_log_norm(mean, scale, scale_matrix, dof):
    """beer/dists/normalwishart.py:219-236 -> [K]."""
    K, D = mean.shape
    idxs = np.arange(1, D + 1, dtype=mean.dtype)
    logdet = _chol_logdet(scale_matrix)
    return (.5 * dof * logdet + .5 * dof * D * math.log(2)
            + .25 * D * (D - 1) * math.log(math.pi)
            + gammaln(.5 * (dof + 1 - idxs)).sum(-1, keepdims=True)
            - .5 * D * np.log(scale) + .5 * D * LOG2PI).sum(-1).astype(mean.dtype)


def nw_natural(mean, scale, scale_matrix, dof):
    """beer/dists/normalwishart.py:242-269 -> [K, D^2+D+2]."""
    K, D = mean.shape
    quad = mean[:, :, None] * mean[:, None, :]
    return np.concatenate([
        scale * mean,
        -.5 * (np.linalg.inv(scale_matrix) + scale[:, :, None] * quad).reshape(K, D * D),
        -.5 * scale.reshape(-1, 1),
        .5 * (dof - D).reshape(-1, 1),
    ], axis=-1).astype(mean.dtype)


def nw_from_natural(eta):
    """beer/dists/normalwishart.py:110-141 -> (mean, scale, scale_matrix, dof)."""
    l = eta.shape[-1] - 2
    D = int(.5 * (-1 + math.sqrt(1 + 4 * l)))
    np1, np2 = eta[:, :D], eta[:, D:D * (D + 1)]
    np3, np4 = eta[:, -2], eta[:, -1]
    scale = -2 * np3
    mean = np1 / scale[:, None]
    quad = mean[:, :, None] * mean[:, None, :]
    W = np.linalg.inv(-2 * np2.reshape(-1, D, D) - scale[:, None, None] * quad)
    dof = 2 * np4 + D
    return mean, scale.reshape(-1, 1), W, dof.reshape(-1, 1)


# ---------------------------------------------------------------------------
# Normal-Gamma (diagonal covariance)
# ---------------------------------------------------------------------------

def ng_expected_stats(mean, scale, shape, rates):
    """beer/dists/normalgamma.py:118-146 -> [K, 2D+2]."""
    D = mean.shape[-1]
    prec = shape / rates
    pqm = (prec * mean ** 2).sum(-1, keepdims=True) + D / scale
    logdet = np.sum(digamma(shape) - np.log(rates), axis=-1, keepdims=True)
    return np.concatenate([prec * mean, prec, pqm, logdet], axis=-1).astype(mean.dtype)


def ng_log_norm(mean, scale, shape, rates):
    """beer/dists/normalgamma.py:151-157 -> [K]."""
    D = rates.shape[-1]
    return (D * gammaln(shape) - shape * np.log(rates).sum(-1, keepdims=True)
            - .5 * D * np.log(scale)).sum(-1).astype(mean.dtype)


def ng_natural(mean, scale, shape, rates):
    """beer/dists/normalgamma.py:163-180 -> [K, 2D+2]."""
    return np.concatenate([scale * mean, -.5 * scale * mean ** 2 - rates,
                           -.5 * scale, shape - .5], axis=-1).astype(mean.dtype)


def ng_from_natural(eta):
    """beer/dists/normalgamma.py:77-94."""
    D = (eta.shape[-1] - 2) // 2
    np1, np2, np3, np4 = eta[:, :D], eta[:, D:2 * D], eta[:, -2], eta[:, -1]
    scale = -2 * np3
    shape = np4 + .5
    mean = np1 / scale[:, None]
    rates = -np2 - .5 * scale[:, None] * mean ** 2
    return mean, scale.reshape(-1, 1), shape.reshape(-1, 1), rates


# ---------------------------------------------------------------------------
# Isotropic Normal-Gamma
# ---------------------------------------------------------------------------

def ing_expected_stats(mean, scale, shape, rate):
    """beer/dists/isonormalgamma.py:119-157 -> [K, D+3]."""
    D = mean.shape[-1]
    prec = shape / rate
    pqm = prec * (mean ** 2).sum(-1, keepdims=True) + D / scale
    logdet = digamma(shape) - np.log(rate)
    return np.concatenate([prec * mean, prec, pqm, logdet], axis=-1).astype(mean.dtype)


def ing_log_norm(mean, scale, shape, rate):
    """beer/dists/isonormalgamma.py:163-168 -> [K]."""
    D = mean.shape[-1]
    return (gammaln(shape) - shape * np.log(rate)
            - .5 * D * np.log(scale)).sum(-1).astype(mean.dtype)


def ing_natural(mean, scale, shape, rate):
    """beer/dists/isonormalgamma.py:174-192 -> [K, D+3]."""
    D = mean.shape[-1]
    return np.concatenate([
        scale * mean,
        -.5 * scale * np.sum(mean ** 2, axis=-1, keepdims=True) - rate,
        -.5 * scale, shape - 1 + .5 * D], axis=-1).astype(mean.dtype)


def ing_from_natural(eta):
    """beer/dists/isonormalgamma.py:78-95."""
    D = eta.shape[-1] - 3
    np1, np2 = eta[:, :D], eta[:, D:D + 1]
    np3, np4 = eta[:, -2].reshape(-1, 1), eta[:, -1].reshape(-1, 1)
    scale = -2 * np3
    shape = np4 + 1 - .5 * D
    mean = np1 / scale
    rate = -np2 - .5 * scale * np.sum(mean * mean, axis=-1, keepdims=True)
    return mean, scale, shape, rate


# ---------------------------------------------------------------------------
# Dirichlet / Gamma
# ---------------------------------------------------------------------------

def dir_expected_stats(conc):
    """beer/dists/dirichlet.py:106-128.  conc [..., d] -> same shape."""
    c = np.atleast_2d(conc)
    out = np.zeros_like(c)
    psi = digamma(c[:, -1])
    out[:, :-1] = digamma(c[:, :-1]) - psi[:, None]
    out[:, -1] = psi - digamma(c.sum(-1))
    return out.reshape(conc.shape).astype(conc.dtype)


def dir_log_norm(conc):
    """beer/dists/dirichlet.py:135-138."""
    return (gammaln(conc).sum(-1) - gammaln(conc.sum(-1))).astype(conc.dtype)


def dir_natural(conc):
    """beer/dists/dirichlet.py:144-159."""
    c = np.atleast_2d(conc)
    out = c - 1
    out[:, -1] = (c - 1).sum(-1)
    return out.reshape(conc.shape).astype(conc.dtype)


def dir_from_natural(eta):
    """beer/dists/dirichlet.py:71-81."""
    e = np.atleast_2d(eta)
    c = e + 1
    c[:, -1] = e[:, -1] - (c - 1)[:, :-1].sum(-1) + 1
    return c.reshape(eta.shape)


def cat_suffstats(data):
    """beer/dists/dirichlet.py:18-21 -- last column <- row sum."""
    out = data.copy().reshape(-1, data.shape[-1])
    out[:, -1] = out.sum(-1)
    return out.reshape(data.shape)


def gamma_expected_stats(shape, rate):
    """beer/dists/gamma.py:112-124."""
    return np.concatenate([shape / rate, digamma(shape) - np.log(rate)], axis=-1)


def gamma_log_norm(shape, rate):
    """beer/dists/gamma.py:129-131."""
    return (gammaln(shape) - shape * np.log(rate)).sum(-1)


def gamma_natural(shape, rate):
    """beer/dists/gamma.py:137-140."""
    return np.concatenate([-rate, shape - 1], axis=-1)


def gamma_from_natural(eta):
    """beer/dists/gamma.py:68-77."""
    e = np.atleast_2d(eta)
    d = e.shape[-1] // 2
    return (e[:, d:] + 1).reshape(-1), (-e[:, :d]).reshape(-1)


def kl_div(exp_stats_q, eta_q, eta_p, lnorm_q, lnorm_p):
    """beer/dists/basedist.py:243-263."""
    return lnorm_p - lnorm_q - np.sum(exp_stats_q * (eta_p - eta_q), axis=-1)


# Family dispatch: std-params tuples <-> the five functions above.
FAMILIES = {
    'full': dict(exp=nw_expected_stats, lnorm=nw_log_norm, nat=nw_natural,
                 from_nat=nw_from_natural,
                 names=('mean', 'scale', 'scale_matrix', 'dof')),
    'diagonal': dict(exp=ng_expected_stats, lnorm=ng_log_norm, nat=ng_natural,
                     from_nat=ng_from_natural,
                     names=('mean', 'scale', 'shape', 'rates')),
    'isotropic': dict(exp=ing_expected_stats, lnorm=ing_log_norm,
                      nat=ing_natural, from_nat=ing_from_natural,
                      names=('mean', 'scale', 'shape', 'rate')),
}


def family_kl(cov_type, post, prior):
    f = FAMILIES[cov_type]
    return kl_div(f['exp'](*post), f['nat'](*post), f['nat'](*prior),
                  f['lnorm'](*post), f['lnorm'](*prior))


def dir_kl(post_conc, prior_conc):
    return kl_div(dir_expected_stats(post_conc), dir_natural(post_conc),
                  dir_natural(prior_conc), dir_log_norm(post_conc),
                  dir_log_norm(prior_conc))


def natural_grad_update(eta_prior, eta_post, stats, lrate):
    """beer/models/parameters.py:134-141."""
    return eta_post + lrate * (eta_prior + stats - eta_post)


# ---------------------------------------------------------------------------
# E-step building blocks
# ---------------------------------------------------------------------------

def normal_llh(stats, exp_T, D):
    """beer/dists/normalwishart.py:88-92 (diag: normalgamma.py:55-59, iso:
    isonormalgamma.py:56-60): stats @ E[T]^T - .5 D ln 2pi  -> [T, K]."""
    return stats @ exp_T.T + stats.dtype.type(-.5 * D * LOG2PI)


def log_weights(conc):
    """beer/models/mixture.py:45-48 -> categorical.py:70-76 ->
    dirichlet.py:18-21,62-64: eye -> suffstats -> stats @ E[T]."""
    d = conc.shape[-1]
    stats = cat_suffstats(np.eye(d, dtype=conc.dtype))
    return stats @ dir_expected_stats(conc)


def log_weights_set(conc):
    """beer/models/mixtureset.py:64-67 (CategoricalSet, conc [S,G]) -> [S,G]."""
    G = conc.shape[-1]
    stats = cat_suffstats(np.eye(G, dtype=conc.dtype))
    return (stats @ dir_expected_stats(conc).T).T


def logsumexp(a, axis):
    """beer/utils.py:105-123 (and torch.logsumexp): -inf safe."""
    with np.errstate(invalid='ignore', divide='ignore'):
        return _sp_logsumexp(a, axis=axis).astype(a.dtype)


def mixture_estep(stats, exp_T, D, lw, labels=None):
    """beer/models/mixture.py:70-93.  Returns (per-frame [T], resps [T,K])."""
    pc = normal_llh(stats, exp_T, D)
    if labels is None:
        w = pc + lw[None]
        lnorm = logsumexp(w, axis=1).reshape(-1, 1)
        log_resps = w - lnorm
        resps = np.exp(log_resps)
        local_kl = np.sum(np.exp(log_resps) * (log_resps - lw[None]), axis=-1)
    else:
        resps = np.zeros((len(stats), len(lw)), dtype=stats.dtype)
        resps[np.arange(len(stats)), labels] = 1
        local_kl = 0.
    return (pc * resps).sum(-1) - local_kl, resps


def mixture_accumulate(stats, resps):
    """beer/models/mixture.py:95-102 -> categorical.py:78-79,
    normalset.py:121-123.  Returns (weights stats [K], normal stats [K,Q])."""
    return cat_suffstats(resps).sum(0), resps.T @ stats


def mixtureset_estep(stats, exp_T, D, lw_set):
    """beer/models/mixtureset.py:85-98.  lw_set [S,G].
    Returns (log_norm [T,S], comp resps [T,S,G])."""
    S, G = lw_set.shape
    pc = normal_llh(stats, exp_T, D).reshape(-1, S, G)
    w = pc + lw_set[None]
    log_norm = logsumexp(w, axis=-1)
    resps = np.exp(w - log_norm[:, :, None])
    return log_norm, resps


def mixtureset_accumulate(stats, comp_resps, state_resps):
    """beer/models/mixtureset.py:100-112.  Returns (weights stats [S,G],
    normal stats [S*G, Q])."""
    T, S, G = comp_resps.shape
    joint = comp_resps * state_resps[:, :, None]
    total = joint.reshape(T, S * G)
    wstats = cat_suffstats(joint.reshape(-1, G)).reshape(T, S, G).sum(0)
    return wstats, total.T @ stats


def gather_states(pc_llh, order):
    """beer/models/modelset.py:140-146."""
    return pc_llh[:, order]


def scatter_states(resps, order, n_total):
    """beer/models/modelset.py:148-154 (repeated ids add)."""
    out = np.zeros((len(resps), n_total), dtype=resps.dtype)
    for i, o in enumerate(order):
        out[:, o] += resps[:, i]
    return out


# ---------------------------------------------------------------------------
# Graph inference  (K9, K10)
# ---------------------------------------------------------------------------

def forward(llhs, init_lp, trans_lp):
    """beer/graph.py:270-278."""
    la = np.full_like(llhs, -np.inf)
    la[0] = llhs[0] + init_lp
    At = trans_lp.T
    for i in range(1, len(llhs)):
        la[i] = llhs[i] + logsumexp(la[i - 1] + At, axis=1)
    return la


def backward(llhs, final_lp, trans_lp):
    """beer/graph.py:280-287."""
    lb = np.full_like(llhs, -np.inf)
    lb[-1] = final_lp
    for i in reversed(range(len(llhs) - 1)):
        lb[i] = logsumexp(trans_lp + llhs[i + 1] + lb[i + 1], axis=1)
    return lb


def posteriors(llhs, init_lp, final_lp, trans_lp, trans_posteriors=False):
    """beer/graph.py:289-326.  Returns (gamma[, xi], lognorm.mean())."""
    la = forward(llhs, init_lp, trans_lp)
    lb = backward(llhs, final_lp, trans_lp)
    lognorm = logsumexp(la + lb, axis=1)
    with np.errstate(invalid='ignore'):
        gamma = np.exp(la + lb - lognorm[:, None])
    if not trans_posteriors:
        return gamma, lognorm.mean()
    S = len(trans_lp)
    with np.errstate(invalid='ignore'):
        log_xi = la[:-1, :, None] + trans_lp[None] + (llhs + lb)[1:, None, :]
        log_xi = log_xi.reshape(-1, S * S)
        lnorm = logsumexp(log_xi, axis=1)
        xi = np.exp(log_xi - lnorm[:, None])
    xi = np.where(xi != xi, np.zeros_like(xi), xi).reshape(-1, S, S)
    return gamma, xi, lognorm.mean()


def best_path(llhs, init_lp, final_lp, trans_lp):
    """beer/graph.py:329-344.  First-index argmax tie-break (torch.argmax)."""
    T, S = llhs.shape
    bt = np.zeros((T, S), dtype=np.int64)
    omega = llhs[0] + init_lp
    At = trans_lp.T
    ar = np.arange(S)
    for i in range(1, T):
        hyp = omega + At
        bt[i] = np.argmax(hyp, axis=1)
        omega = llhs[i] + hyp[ar, bt[i]]
    path = [int(np.argmax(omega + final_lp))]
    for i in reversed(range(1, T)):
        path.insert(0, int(bt[i, path[0]]))
    return np.asarray(path, dtype=np.int64)


def onehot(labels, n, dtype):
    """beer/utils.py:84-102."""
    out = np.zeros((len(labels), n), dtype=dtype)
    out[np.arange(len(labels)), labels] = 1
    return out


def hmm_estep(pc_llhs_all, order, init_lp, final_lp, trans_lp, scale=1.,
              viterbi=False, state_path=None, trans_posteriors=False):
    """beer/models/hmm.py:40-92.  `pc_llhs_all` [T, S_total] are per-pdf
    log-likelihoods (NormalSet llh or MixtureSet log_norm).  Returns dict with
    pc_llhs [T,S_u], resps, optional trans_resps, exp_llh [T]."""
    dtype = pc_llhs_all.dtype
    pc = dtype.type(scale) * gather_states(pc_llhs_all, order)
    out = {'pc_llhs': pc}
    if viterbi or state_path is not None:
        path = best_path(pc, init_lp, final_lp, trans_lp) \
            if state_path is None else np.asarray(state_path)
        out['path'] = path
        resps = onehot(path, len(trans_lp), dtype)
        if trans_posteriors:
            S = len(trans_lp)
            xi = np.zeros((len(pc) - 1, S, S), dtype=np.float32)
            xi[np.arange(len(pc) - 1), path[:-1], path[1:]] = 1
            out['trans_resps'] = xi
    elif trans_posteriors:
        resps, xi, _ = posteriors(pc, init_lp, final_lp, trans_lp, True)
        out['trans_resps'] = xi
    else:
        resps, _ = posteriors(pc, init_lp, final_lp, trans_lp, False)
    out['resps'] = resps
    out['exp_llh'] = (pc * resps).sum(-1)
    return out


def phone_counts(trans_resps, resps, start_idxs, end_idxs):
    """beer/models/phoneloop.py:88-95 (before the categorical suffstats)."""
    tr = trans_resps.sum(0)
    pr = tr[:, start_idxs][end_idxs, :].sum(0)
    return pr + resps[0][start_idxs]


def phoneloop_update_trans(trans_lp, log_w, start_idxs, end_idxs):
    """beer/models/phoneloop.py:53-65 (in place, same visiting order)."""
    for e in end_idxs:
        loop_prob = np.exp(trans_lp[e, e])
        trans_lp[e, start_idxs] = np.log(1 - loop_prob) + log_w
    return trans_lp


# ---------------------------------------------------------------------------
# Stick-breaking categorical  (a15)
# ---------------------------------------------------------------------------

def reverse_ordering(ordering):
    """beer/models/categorical.py:133-138."""
    rev = np.zeros_like(ordering)
    rev[ordering] = np.arange(len(ordering))
    return rev


def sb_transform_stats(stats):
    """beer/models/categorical.py:109-118.  stats [P] -> (ordering, [P,2])."""
    ordering = np.argsort(-stats, kind='stable')
    s = stats[ordering]
    s2 = np.zeros_like(s)
    s2[:-1] = s[1:]
    s2 = np.flip(np.cumsum(np.flip(s2)))
    new = np.stack([s, s2], axis=-1)
    new[:, -1] += new[:, :-1].sum(-1)
    return ordering, new[reverse_ordering(ordering)]


def sb_log_prob(conc, ordering):
    """beer/models/categorical.py:120-131.  Returns (log_prob, log_1_v) in
    the sorted order."""
    c = conc[ordering]
    s_dig = digamma(c.sum(-1))
    log_v = digamma(c[:, 0]) - s_dig
    log_1_v = digamma(c[:, 1]) - s_dig
    lp = log_v.copy()
    lp[1:] += np.cumsum(log_1_v[:-1])
    return lp, log_1_v


def sb_log_weights(conc, ordering):
    """beer/models/categorical.py:157-159 with stats = eye."""
    lp, _ = sb_log_prob(conc, ordering)
    return lp[reverse_ordering(ordering)]


# ---------------------------------------------------------------------------
# Whole-model steps (used by tests and by bench.py's cpu_baseline leg)
# ---------------------------------------------------------------------------

def gmm_elbo_step(X, cov_type, post, prior, w_post, w_prior, datasize=-1,
                  labels=None):
    """One `evidence_lower_bound(Mixture, X)` call, reference-faithful op
    sequence: beer/inference/objectives.py:175-190 driving mixture.py:67-102.
    Returns dict(value, kl, per_frame, resps, acc_normal [K,Q], acc_weights [K])."""
    f = FAMILIES[cov_type]
    D = X.shape[1]
    T = len(X)
    if datasize <= 0:
        datasize = T
    scale = datasize / float(T)
    stats = SUFFSTATS[cov_type](X)                       # K1
    exp_T = f['exp'](*post)                              # K2
    lw = log_weights(w_post)                             # K4
    per_frame, resps = mixture_estep(stats, exp_T, D, lw, labels)   # K3,K5,K6
    kl = family_kl(cov_type, post, prior).sum() + dir_kl(w_post, w_prior).sum()  # K13
    value = float(scale) * per_frame.sum() - kl
    acc_w, acc_n = mixture_accumulate(stats, resps)      # K7
    return dict(value=value, kl=kl, per_frame=per_frame, resps=resps,
                acc_normal=acc_n, acc_weights=acc_w, scale=scale)


def gmm_mstep(cov_type, post, prior, w_post, w_prior, acc_normal, acc_weights,
              lrate=1.):
    """`elbo.backward(); optim.step()` for a Mixture: objectives.py:92-107,
    optimizers.py:27-31, parameters.py:134-141.  `acc_*` are already scaled by
    datasize / minibatchsize.  Returns (new post tuple, new w_post)."""
    f = FAMILIES[cov_type]
    eta = natural_grad_update(f['nat'](*prior), f['nat'](*post), acc_normal, lrate)
    new_post = f['from_nat'](eta)
    eta_w = natural_grad_update(dir_natural(w_prior), dir_natural(w_post),
                                acc_weights, lrate)
    return new_post, dir_from_natural(eta_w)


# ---------------------------------------------------------------------------
# HMM / PhoneLoop whole-model steps
# ---------------------------------------------------------------------------
#
# An "emission group" is one entry of a JointModelSet (modelset.py:43-109):
#   dict(cov_type=..., post=(std params), prior=(std params),
#        S=#pdfs, G=#components per pdf (0 => plain NormalSet, one Normal per
#        pdf), w_post=[S,G], w_prior=[S,G])

def emissions_estep(X, groups, stats_scale=None):
    """JointModelSet.expected_log_likelihood (modelset.py:71-75) over
    MixtureSet (mixtureset.py:85-98) or NormalSet (normalset.py:117-119)
    groups.  Returns (pc_llh_all [T, S_total], per-group cache).
    `stats_scale` reproduces HMM.posteriors (hmm.py:116-121), which scales the
    *statistics* (not the log-likelihoods) by the acoustic scale."""
    D = X.shape[1]
    cols, cache = [], []
    for g in groups:
        stats = SUFFSTATS[g['cov_type']](X)
        if stats_scale is not None:
            stats = stats * stats.dtype.type(stats_scale)
        exp_T = FAMILIES[g['cov_type']]['exp'](*g['post'])
        if g['G'] == 0:
            cols.append(normal_llh(stats, exp_T, D))
            cache.append((stats, None))
        else:
            ln, cr = mixtureset_estep(stats, exp_T, D, log_weights_set(g['w_post']))
            cols.append(ln)
            cache.append((stats, cr))
    return np.concatenate(cols, axis=-1), cache


def emissions_accumulate(groups, cache, resps_all):
    """JointModelSet.accumulate (modelset.py:77-85).  Returns per group
    (normal stats [K,Q], weights stats [S,G] or None)."""
    out, start = [], 0
    for g, (stats, cr) in zip(groups, cache):
        r = resps_all[:, start:start + g['S']]
        start += g['S']
        if g['G'] == 0:
            out.append((r.T @ stats, None))
        else:
            ws, ns = mixtureset_accumulate(stats, cr, r)
            out.append((ns, ws))
    return out


def emissions_kl(groups):
    kl = 0.
    for g in groups:
        kl = kl + family_kl(g['cov_type'], g['post'], g['prior']).sum()
        if g['G'] > 0:
            kl = kl + dir_kl(g['w_post'], g['w_prior']).sum()
    return kl


def hmm_elbo_step(X, groups, graph, datasize=-1, scale=1., viterbi=False,
                  state_path=None, trans_posteriors=False, extra_kl=0.):
    """`evidence_lower_bound(hmm, X, ...)`: objectives.py:175-190 driving
    hmm.py:73-100.  `graph` = dict(init, final, trans, order).  Returns dict
    with value, exp_llh, resps (+trans_resps), acc (per group), kl."""
    T = len(X)
    if datasize <= 0:
        datasize = T
    pc_all, cache = emissions_estep(X, groups)
    r = hmm_estep(pc_all, graph['order'], graph['init'], graph['final'],
                  graph['trans'], scale=scale, viterbi=viterbi,
                  state_path=state_path, trans_posteriors=trans_posteriors)
    kl = emissions_kl(groups) + extra_kl
    value = float(datasize / float(T)) * r['exp_llh'].sum() - kl
    scaled = pc_all.dtype.type(scale) * r['resps']            # hmm.py:95
    resps_all = scatter_states(scaled, graph['order'], pc_all.shape[1])
    r.update(value=value, kl=kl, pc_all=pc_all,
             acc=emissions_accumulate(groups, cache, resps_all))
    return r


# ---------------------------------------------------------------------------
# The prior of a VAE (beer/models/vae.py:63-86): value and gradient w.r.t. the samples
# ---------------------------------------------------------------------------

def suffstats_backward(cov_type, X, grad_stats):
    """What autograd gives the reference for `sufficient_statistics` (normalwishart.py:30-38,
    normalgamma.py:20-27, isonormalgamma.py:21-30): J_phi(x_t)^T grad_stats[t] -> [T, D]."""
    T, D = X.shape
    out = grad_stats[:, :D].copy()
    if cov_type == 'full':
        G = grad_stats[:, D:D + D * D].reshape(T, D, D)
        out -= .5 * (np.einsum('tij,tj->ti', G, X) + np.einsum('tji,tj->ti', G, X))
    elif cov_type == 'diagonal':
        out -= grad_stats[:, D:2 * D] * X
    else:
        out -= grad_stats[:, D:D + 1] * X
    return out


def prior_gradient_wrt_samples(cov_type, Z, weights, exp_T, upstream=None):
    """d/dZ of sum_t upstream[t] * sum_k weights[t,k] * (phi(z_t) . exp_T[k]) with `weights`
    detached: the reference's autograd through `(pc_llhs * resps).sum(-1)` (mixture.py:79,92;
    hmm.py:81-87), `stats @ exp_T^T` (normalwishart.py:88-92) and `sufficient_statistics`,
    for ONE sample per frame (vae.py:73-74: the mean over one sample is the sample's phi)."""
    w = weights if upstream is None else weights * upstream[:, None]
    return suffstats_backward(cov_type, Z, w @ exp_T)


def vae_gmm_prior(cov_type, Z, post, w_post):
    """`Mixture.expected_log_likelihood(phi(Z))` as a VAE's prior with one sample per frame
    (mixture.py:70-93): (per-frame value [T], responsibilities [T, K], E[T] [K, Q])."""
    exp_T = FAMILIES[cov_type]['exp'](*post)
    value, resps = mixture_estep(SUFFSTATS[cov_type](Z), exp_T, Z.shape[1], log_weights(w_post))
    return value, resps, exp_T


def vae_hmm_prior(cov_type, Z, post, graph):
    """`HMM.expected_log_likelihood(phi(Z))` likewise (hmm.py:73-92): (per-frame value [T],
    state posteriors scattered to the pdf ids [T, S_total], E[T])."""
    exp_T = FAMILIES[cov_type]['exp'](*post)
    pc_all = normal_llh(SUFFSTATS[cov_type](Z), exp_T, Z.shape[1])
    r = hmm_estep(pc_all, graph['order'], graph['init'], graph['final'], graph['trans'])
    return r['exp_llh'], scatter_states(r['resps'], graph['order'], pc_all.shape[1]), exp_T


def emissions_mstep(groups, acc, scale, lrate=1.):
    """backward + step on every emission parameter (objectives.py:98-105,
    parameters.py:134-141).  Returns new groups."""
    new = []
    for g, (ns, ws) in zip(groups, acc):
        f = FAMILIES[g['cov_type']]
        eta = natural_grad_update(f['nat'](*g['prior']), f['nat'](*g['post']),
                                  scale * ns, lrate)
        ng = dict(g, post=f['from_nat'](eta))
        if g['G'] > 0:
            eta_w = natural_grad_update(dir_natural(g['w_prior']),
                                        dir_natural(g['w_post']), scale * ws, lrate)
            ng['w_post'] = dir_from_natural(eta_w)
        new.append(ng)
    return new


def categorical_mstep(kind, state, stats, lrate=1.):
    """M-step of the phone-weights model.  kind 'dirichlet':
    categorical.py:39-79; 'dirichlet_process': categorical.py:82-163;
    'gamma_dirichlet_process': categorical.py:166-209.  `state` is a dict with
    post/prior concentrations (+ ordering, + gamma shape/rate); `stats` are
    the stored (already scaled) statistics [P].  Returns the new state."""
    st = dict(state)
    if kind == 'dirichlet':
        eta = natural_grad_update(dir_natural(st['prior']), dir_natural(st['post']),
                                  stats, lrate)
        st['post'] = dir_from_natural(eta)
        return st
    ordering, tstats = sb_transform_stats(stats)              # before-update callback
    st['ordering'] = ordering
    eta = natural_grad_update(dir_natural(st['prior']), dir_natural(st['post']),
                              tstats, lrate)
    st['post'] = dir_from_natural(eta)
    if kind == 'gamma_dirichlet_process':                     # after-update callback
        _, log_1_v = sb_log_prob(st['post'], ordering)
        g_stats = np.array([log_1_v.sum(), float(len(log_1_v))])
        eta_g = natural_grad_update(
            gamma_natural(st['g_prior_shape'], st['g_prior_rate']),
            gamma_natural(st['g_post_shape'], st['g_post_rate']), g_stats, 1.)
        st['g_post_shape'], st['g_post_rate'] = gamma_from_natural(eta_g)
        prior = st['prior'].copy()
        prior[:, 1] = st['g_post_shape'] / st['g_post_rate']
        st['prior'] = prior
    return st


def categorical_log_weights(kind, state):
    if kind == 'dirichlet':
        return log_weights(state['post'])
    return sb_log_weights(state['post'], state['ordering'])


def categorical_kl(kind, state):
    return dir_kl(state['post'], state['prior']).sum()
