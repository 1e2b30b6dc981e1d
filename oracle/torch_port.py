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
