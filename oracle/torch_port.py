"""CPU timing baseline: the GMM VB iteration in the reference's own operation
sequence, on torch CPU tensors.

TEST INFRASTRUCTURE ONLY (same rule as beer_oracle.py: only tests/, smoke()
and bench.py's cpu_baseline leg may import this).

Why a second restatement: the numpy oracle is single-threaded in its
element-wise passes, while the reference runs on torch, whose `mul` / `cat` /
`exp` are multi-threaded -- and those passes are 68 % of the reference's GMM
step.  To report "beer's own CPU vbi path timed
on the host cores" without shipping the reference's files, this module replays
the reference's op sequence with the same torch ops:

    materialise phi(X) with mul + cat            normalwishart.py:30-38
    E[T] with cholesky / digamma                 normalwishart.py:170-210
    stats @ E[T]^T                               normalwishart.py:88-92
    logsumexp, exp, local KL                     mixture.py:79-93
    KL(q || p): inverse, cholesky, lgamma        basedist.py:243-263, normalwishart.py:219-269
    resps^T @ stats                              normalset.py:121-123
    natural-gradient step + inverse              parameters.py:134-141, normalwishart.py:110-141

It is pinned twice: numerically against the numpy oracle (tests/
test_oracle_golden.py::test_torch_port_matches_oracle) and in wall-clock
against the imported reference on the build container's 8 vCPUs.
"""

import math

import torch

LOG2PI = math.log(2 * math.pi)


def nw_exp_stats(mean, scale, W, dof):
    K, D = mean.shape
    idx = torch.arange(1, D + 1, dtype=mean.dtype)
    L = torch.linalg.cholesky(W)
    logdet = 2 * torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1, keepdim=True)
    prec = dof[:, :, None] * W
    mq = mean[:, :, None] * mean[:, None, :]
    return torch.cat([
        torch.matmul(prec, mean[:, :, None]).reshape(K, D), prec.reshape(K, D * D),
        D / scale + (prec.reshape(K, -1) * mq.reshape(K, -1)).sum(-1, keepdim=True),
        torch.digamma(.5 * (dof + 1 - idx)).sum(-1, keepdim=True) + D * math.log(2) + logdet,
    ], dim=-1)


def nw_log_norm(mean, scale, W, dof):
    K, D = mean.shape
    idx = torch.arange(1, D + 1, dtype=mean.dtype)
    L = torch.linalg.cholesky(W)
    logdet = 2 * torch.log(torch.diagonal(L, dim1=-2, dim2=-1)).sum(-1, keepdim=True)
    return (.5 * dof * logdet + .5 * dof * D * math.log(2)
            + .25 * D * (D - 1) * math.log(math.pi)
            + torch.lgamma(.5 * (dof + 1 - idx)).sum(-1, keepdim=True)
            - .5 * D * torch.log(scale) + .5 * D * LOG2PI).sum(-1)


def nw_natural(mean, scale, W, dof):
    K, D = mean.shape
    mq = mean[:, :, None] * mean[:, None, :]
    return torch.cat([scale * mean,
                      -.5 * (torch.linalg.inv(W) + scale[:, :, None] * mq).reshape(K, D * D),
                      -.5 * scale.reshape(-1, 1), .5 * (dof - D).reshape(-1, 1)], dim=-1)


def nw_from_natural(eta, D):
    scale = -2 * eta[:, -2]
    mean = eta[:, :D] / scale[:, None]
    mq = mean[:, :, None] * mean[:, None, :]
    W = torch.linalg.inv(-2 * eta[:, D:D * (D + 1)].reshape(-1, D, D) - scale[:, None, None] * mq)
    return mean, scale.reshape(-1, 1), W, (2 * eta[:, -1] + D).reshape(-1, 1)


def dir_exp_stats(c):
    out = torch.zeros_like(c)
    psi = torch.digamma(c[-1])
    out[:-1] = torch.digamma(c[:-1]) - psi
    out[-1] = psi - torch.digamma(c.sum())
    return out


def dir_natural(c):
    out = c - 1
    out[-1] = (c - 1).sum()
    return out


def dir_log_norm(c):
    return torch.lgamma(c).sum() - torch.lgamma(c.sum())


def gmm_elbo(X, post, prior, w_post, w_prior, datasize):
    'One `evidence_lower_bound(Mixture, X)` call, full covariance.'
    T, D = X.shape
    one = torch.ones(T, 1, dtype=X.dtype)
    quad = X[:, :, None] * X[:, None, :]
    stats = torch.cat([X, -.5 * quad.reshape(T, -1), -.5 * one, .5 * one], dim=-1)
    exp_T = nw_exp_stats(*post)
    pc = stats @ exp_T.t() - .5 * D * LOG2PI
    eye = torch.eye(len(w_post), dtype=X.dtype)
    eye[:, -1] = eye.sum(-1)
    lw = eye @ dir_exp_stats(w_post)
    w = pc + lw[None]
    lnorm = torch.logsumexp(w, dim=1).view(-1, 1)
    log_r = w - lnorm
    resps = log_r.exp()
    local_kl = torch.sum(log_r.exp() * (log_r - lw[None]), dim=-1)
    per_frame = (pc * resps).sum(-1) - local_kl
    kl = (nw_log_norm(*prior) - nw_log_norm(*post)
          - torch.sum(exp_T * (nw_natural(*prior) - nw_natural(*post)), dim=-1)).sum()
    kl = kl + dir_log_norm(w_prior) - dir_log_norm(w_post) \
        - torch.sum(dir_exp_stats(w_post) * (dir_natural(w_prior) - dir_natural(w_post)))
    value = (datasize / float(T)) * per_frame.sum() - kl
    rs = resps.clone()
    rs[:, -1] = rs.sum(-1)
    return value, resps.t() @ stats, rs.sum(0)


def gmm_update(post, prior, w_post, w_prior, acc_n, acc_w, D):
    eta = nw_natural(*post)
    eta = eta + (nw_natural(*prior) + acc_n - eta)
    eta_w = dir_natural(w_post)
    eta_w = eta_w + (dir_natural(w_prior) + acc_w - eta_w)
    c = eta_w + 1
    c[-1] = eta_w[-1] - eta_w[:-1].sum() + 1
    return nw_from_natural(eta, D), c


def gmm_iteration(X, post, prior, w_post, w_prior, chunk):
    '''E-step over `chunk`-frame utterances + M-step, as `beer hmm accumulate`
    / `update` would drive it.  Returns (summed ELBO value, new posterior).'''
    N, D = X.shape
    total, acc_n, acc_w = 0., 0., 0.
    for lo in range(0, N, chunk):
        v, an, aw = gmm_elbo(X[lo:lo + chunk], post, prior, w_post, w_prior, N)
        total, acc_n, acc_w = total + v, acc_n + an, acc_w + aw
    new_post, new_w = gmm_update(post, prior, w_post, w_prior, acc_n, acc_w, D)
    return float(total), new_post, new_w


# ---------------------------------------------------------------------------
# Phone-loop HMM with diagonal-covariance mixture emissions (BASELINE config 3):
# one `evidence_lower_bound(PhoneLoop, utterance)` call in the reference's op
# sequence --
#     phi(X) with mul + cat                                normalgamma.py:20-27
#     E[T] with digamma / log                              normalgamma.py:118-146
#     stats @ E[T]^T, per-state logsumexp, resps           mixtureset.py:85-98
#     gather by pdf id, scale                              modelset.py:140-146, hmm.py:79
#     forward / backward: a Python loop over frames of
#       torch.logsumexp over the dense [S, S] matrix       graph.py:270-287
#     posteriors + [T-1, S, S] transition posteriors       graph.py:289-326
#     joint responsibilities, resps^T @ stats              mixtureset.py:100-112
#     KL(q || p) of every parameter, once per utterance    objectives.py:183
# Numerically pinned against the numpy oracle
# (tests/test_oracle_golden.py::test_torch_port_hmm_matches_oracle).
# ---------------------------------------------------------------------------

def ng_exp_stats(mean, scale, shape, rates):
    D = mean.shape[-1]
    prec = shape / rates
    pqm = (prec * mean ** 2).sum(-1, keepdim=True) + D / scale
    logdet = (torch.digamma(shape) - torch.log(rates)).sum(-1, keepdim=True)
    return torch.cat([prec * mean, prec, pqm, logdet], dim=-1)


def ng_log_norm(mean, scale, shape, rates):
    D = rates.shape[-1]
    return (D * torch.lgamma(shape) - shape * torch.log(rates).sum(-1, keepdim=True)
            - .5 * D * torch.log(scale)).sum(-1)


def ng_natural(mean, scale, shape, rates):
    return torch.cat([scale * mean, -.5 * scale * mean ** 2 - rates, -.5 * scale, shape - .5],
                     dim=-1)


def ng_from_natural(eta, D):
    'normalgamma.py:77-94.'
    scale = -2 * eta[:, 2 * D:2 * D + 1]
    shape = eta[:, 2 * D + 1:] + .5
    mean = eta[:, :D] / scale
    rates = -eta[:, D:2 * D] - .5 * scale * mean ** 2
    return mean, scale, shape, rates


def gmm_diag_elbo(X, post, prior, w_post, w_prior, datasize):
    '''One `evidence_lower_bound(Mixture, X)` call with diagonal covariances (BASELINE
    config 1: examples/Mixture Model.ipynb): phi(X) with mul + cat (normalgamma.py:20-27),
    E[T] (118-146), stats @ E[T]^T (55-59), mixture.py:70-102, KL per call
    (basedist.py:243-263, normalgamma.py:151-157).'''
    T, D = X.shape
    one = torch.ones(T, 1, dtype=X.dtype)
    stats = torch.cat([X, -.5 * X ** 2, -.5 * one, .5 * one], dim=-1)
    exp_T = ng_exp_stats(*post)
    pc = stats @ exp_T.t() - .5 * D * LOG2PI
    eye = torch.eye(len(w_post), dtype=X.dtype)
    eye[:, -1] = eye.sum(-1)
    lw = eye @ dir_exp_stats(w_post)
    w = pc + lw[None]
    lnorm = torch.logsumexp(w, dim=1).view(-1, 1)
    log_r = w - lnorm
    resps = log_r.exp()
    local_kl = torch.sum(log_r.exp() * (log_r - lw[None]), dim=-1)
    per_frame = (pc * resps).sum(-1) - local_kl
    kl = (ng_log_norm(*prior) - ng_log_norm(*post)
          - torch.sum(exp_T * (ng_natural(*prior) - ng_natural(*post)), dim=-1)).sum()
    kl = kl + dir_log_norm(w_prior) - dir_log_norm(w_post) \
        - torch.sum(dir_exp_stats(w_post) * (dir_natural(w_prior) - dir_natural(w_post)))
    value = (datasize / float(T)) * per_frame.sum() - kl
    rs = resps.clone()
    rs[:, -1] = rs.sum(-1)
    return value, resps.t() @ stats, rs.sum(0)


def gmm_diag_iteration(X, post, prior, w_post, w_prior):
    '''The loop body of examples/Mixture Model.ipynb cell 9 (init_step, evidence_lower_bound,
    backward, step): (ELBO value, new posterior, new weight concentrations).'''
    N, D = X.shape
    value, acc_n, acc_w = gmm_diag_elbo(X, post, prior, w_post, w_prior, N)
    eta = ng_natural(*post)
    eta = eta + (ng_natural(*prior) + acc_n - eta)
    eta_w = dir_natural(w_post)
    eta_w = eta_w + (dir_natural(w_prior) + acc_w - eta_w)
    c = eta_w + 1
    c[-1] = eta_w[-1] - eta_w[:-1].sum() + 1
    return float(value), ng_from_natural(eta, D), c


def dirset_exp_stats(c):
    'Rows of Dirichlet concentrations [S, G] -> E[T] [S, G] (dirichlet.py:106-128).'
    out = torch.zeros_like(c)
    psi = torch.digamma(c[:, -1])
    out[:, :-1] = torch.digamma(c[:, :-1]) - psi[:, None]
    out[:, -1] = psi - torch.digamma(c.sum(-1))
    return out


def dirset_natural(c):
    out = c - 1
    out[:, -1] = (c - 1).sum(-1)
    return out


def dirset_log_norm(c):
    return torch.lgamma(c).sum(-1) - torch.lgamma(c.sum(-1))


def hmm_forward_backward(llhs, init_lp, final_lp, trans_lp):
    'graph.py:270-326: (gamma [T,S], xi [T-1,S,S]).'
    T, S = llhs.shape
    la = torch.full_like(llhs, -float('inf'))
    la[0] = llhs[0] + init_lp
    At = trans_lp.t()
    for i in range(1, T):
        la[i] = llhs[i] + torch.logsumexp(la[i - 1] + At, dim=1)
    lb = torch.full_like(llhs, -float('inf'))
    lb[-1] = final_lp
    for i in reversed(range(T - 1)):
        lb[i] = torch.logsumexp(trans_lp + llhs[i + 1] + lb[i + 1], dim=1)
    lognorm = torch.logsumexp(la + lb, dim=1)
    gamma = torch.exp(la + lb - lognorm[:, None])
    log_xi = la[:-1, :, None] + trans_lp[None] + (llhs + lb)[1:, None, :]
    log_xi = log_xi.reshape(-1, S * S)
    xi = torch.exp(log_xi - torch.logsumexp(log_xi, dim=1)[:, None])
    xi[xi != xi] = 0.
    return gamma, xi.reshape(-1, S, S)


def hmm_elbo(X, post, prior, w_post, w_prior, init_lp, final_lp, trans_lp, datasize,
             trans_posteriors=True, order=None):
    '''One utterance through a phone-loop HMM whose S states each have a G-component
    diagonal mixture (pdf ids = states).  post / prior: Normal-Gamma std params of
    the S*G Gaussians; w_post / w_prior [S, G].  Returns (value, Gaussian stats
    [S*G, 2D+2], weight stats [S, G], summed transition posteriors [S, S]).
    `order` (an inference graph's `pdf_id_mapping`, e.g. an alignment graph: `beer hmm
    accumulate --alis`): the graph's states read the per-pdf log-likelihoods through it
    (modelset.py:140-146) and their posteriors are added back column by column in a Python
    loop (modelset.py:148-154), as the reference does.'''
    value, accs, xi = hmm_elbo_groups(X, [(post, prior, w_post, w_prior)], init_lp, final_lp,
                                      trans_lp, datasize, trans_posteriors, order)
    return value, accs[0][0], accs[0][1], xi


def hmm_elbo_groups(X, groups, init_lp, final_lp, trans_lp, datasize, trans_posteriors=True,
                    order=None):
    '''`hmm_elbo` for emissions that are a JointModelSet of several MixtureSets (the recipes'
    models: one group per entry of conf/hmm.yml, mkphones.py:100-113): `groups` =
    [(post, prior, w_post [S_g, G_g], w_prior)], pdf ids running through the groups in order
    (modelset.py:71-85: per-group log-likelihoods concatenated, posteriors split back).
    Returns (value, [(Gaussian stats, weight stats) per group], summed transition posteriors).'''
    T, D = X.shape
    one = torch.ones(T, 1, dtype=X.dtype)
    stats = torch.cat([X, -.5 * X ** 2, -.5 * one, .5 * one], dim=-1)
    cols, comps, kl = [], [], 0.
    for post, prior, w_post, w_prior in groups:
        S, G = w_post.shape
        exp_T = ng_exp_stats(*post)
        pc = (stats @ exp_T.t() - .5 * D * LOG2PI).reshape(T, S, G)
        eye = torch.eye(G, dtype=X.dtype)
        eye[:, -1] = eye.sum(-1)
        lw = (eye @ dirset_exp_stats(w_post).t()).t()
        w = pc + lw[None]
        ln = torch.logsumexp(w, dim=-1)
        cols.append(ln)
        comps.append(torch.exp(w - ln[:, :, None]))
        kl = kl + (ng_log_norm(*prior) - ng_log_norm(*post)
                   - torch.sum(exp_T * (ng_natural(*prior) - ng_natural(*post)), dim=-1)).sum()
        kl = kl + (dirset_log_norm(w_prior) - dirset_log_norm(w_post)
                   - torch.sum(dirset_exp_stats(w_post) * (dirset_natural(w_prior)
                                                           - dirset_natural(w_post)), dim=-1)).sum()
    log_norm = cols[0] if len(cols) == 1 else torch.cat(cols, dim=-1)
    if order is None:
        gamma, xi = hmm_forward_backward(log_norm, init_lp, final_lp, trans_lp)
        exp_llh = (log_norm * gamma).sum(-1)
    else:
        idx = torch.as_tensor(order, dtype=torch.long)
        pc_llhs = log_norm[:, idx]
        g_u, xi = hmm_forward_backward(pc_llhs, init_lp, final_lp, trans_lp)
        exp_llh = (pc_llhs * g_u).sum(-1)
        gamma = torch.zeros_like(log_norm)
        for i, pdf_id in enumerate(order):
            gamma[:, pdf_id] += g_u[:, i]
    value = (datasize / float(T)) * exp_llh.sum() - kl
    accs, first = [], 0
    for (post, prior, w_post, w_prior), comp in zip(groups, comps):
        S, G = w_post.shape
        joint = comp * gamma[:, first:first + S, None]
        first += S
        wstats = joint.reshape(-1, G).clone()
        wstats[:, -1] = wstats.sum(-1)
        wstats = wstats.reshape(T, S, G).sum(0)
        accs.append((joint.reshape(T, S * G).t() @ stats, wstats))
    return value, accs, (xi.sum(0) if trans_posteriors else None)


# ---------------------------------------------------------------------------
# The prior of an HMM-VAE (BASELINE config 4): what `VAE.expected_log_likelihood` asks of its
# HMM prior for one utterance with one sample per frame (vae.py:63-86, hmm.py:73-100) --
#     phi(z) with mul + cat                      normalgamma.py:20-27 / normalwishart.py:30-38
#     stats @ E[T]^T                             normalset.py:117-119
#     the Python forward-backward loop           graph.py:270-326   (on detached values, hmm.py:80)
#     sum_s gamma l, autograd back to z          hmm.py:87, torch.autograd
#     gamma^T @ stats                            normalset.py:121-123
# Pinned on the numpy oracle's `vae_hmm_prior` / `prior_gradient_wrt_samples`
# (tests/test_oracle_golden.py::test_torch_port_vae_prior_matches_oracle), which are pinned on
# the reference's G18 goldens.
# ---------------------------------------------------------------------------

def vae_hmm_prior_path(z, cov_type, post, init_lp, final_lp, trans_lp):
    """(per-frame value [T], d sum(value) / dz [T, D], accumulated statistics [S, Q]) of an HMM
    prior with ONE Gaussian per state over the latent samples `z` [T, D]."""
    z = z.detach().clone().requires_grad_(True)
    T, D = z.shape
    one = torch.ones(T, 1, dtype=z.dtype)
    if cov_type == 'full':
        quad = (z[:, :, None] * z[:, None, :]).reshape(T, -1)
        exp_T = nw_exp_stats(*post)
    else:
        quad = z ** 2
        exp_T = ng_exp_stats(*post)
    stats = torch.cat([z, -.5 * quad, -.5 * one, .5 * one], dim=-1)
    pc = stats @ exp_T.t() - .5 * D * LOG2PI
    gamma, _ = hmm_forward_backward(pc.detach(), init_lp, final_lp, trans_lp)
    value = (pc * gamma).sum(-1)
    value.sum().backward()
    return value.detach(), z.grad, gamma.t() @ stats.detach()
