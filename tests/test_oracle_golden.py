"""Pin the CPU oracle (`oracle/beer_oracle.py`) against golden vectors produced
by importing the reference (tests/golden/make_golden.py).  CPU only."""

import numpy as np
import pytest

from helpers import (COV_OF, assert_close, dist_cls, load_golden, n_params,
                     orc, std_params)

TOL64 = 1e-10
TOL32 = 2e-4      # fp32 goldens: the reference's own fp32 rounding band


# --- G10: distribution level -------------------------------------------------

@pytest.mark.parametrize('name,fam', [('nw', 'full'), ('ng', 'diagonal'),
                                      ('ing', 'isotropic')])
def test_g10_normal_families(name, fam):
    g = load_golden('g10_dists')
    f = orc.FAMILIES[fam]
    q, p = std_params(g, f'{name}.q'), std_params(g, f'{name}.p')
    assert_close(f['nat'](*q), g[f'{name}.natural'], TOL64, 'natural')
    assert_close(f['exp'](*q), g[f'{name}.exp_stats'], TOL64, 'E[T]')
    assert_close(f['lnorm'](*q), g[f'{name}.log_norm'], TOL64, 'log_norm')
    assert_close(orc.family_kl(fam, q, p), g[f'{name}.kl'], 1e-9, 'kl')
    rt = f['from_nat'](f['nat'](*q))
    for arr, pn in zip(rt, f['names']):
        assert_close(arr.reshape(g[f'{name}.roundtrip.{pn}'].shape),
                     g[f'{name}.roundtrip.{pn}'], 1e-9, 'roundtrip ' + pn)


@pytest.mark.parametrize('name', ['dir', 'dirset'])
def test_g10_dirichlet(name):
    g = load_golden('g10_dists')
    (q,), (p,) = std_params(g, f'{name}.q'), std_params(g, f'{name}.p')
    assert_close(orc.dir_natural(q), g[f'{name}.natural'], TOL64)
    assert_close(orc.dir_expected_stats(q), g[f'{name}.exp_stats'], TOL64)
    assert_close(orc.dir_log_norm(q), g[f'{name}.log_norm'], TOL64)
    assert_close(orc.dir_kl(q, p), g[f'{name}.kl'], 1e-9)
    assert_close(orc.dir_from_natural(orc.dir_natural(q)),
                 g[f'{name}.roundtrip.concentrations'], TOL64)


def test_g10_gamma_and_stats():
    g = load_golden('g10_dists')
    q, p = std_params(g, 'gamma.q'), std_params(g, 'gamma.p')
    assert_close(orc.gamma_natural(*q), g['gamma.natural'], TOL64)
    assert_close(orc.gamma_expected_stats(*q), g['gamma.exp_stats'], TOL64)
    assert_close(orc.gamma_log_norm(*q), g['gamma.log_norm'], TOL64)
    kl = orc.kl_div(orc.gamma_expected_stats(*q), orc.gamma_natural(*q),
                    orc.gamma_natural(*p), orc.gamma_log_norm(*q),
                    orc.gamma_log_norm(*p))
    assert_close(kl, g['gamma.kl'], 1e-9)
    for cov in ('full', 'diagonal', 'isotropic'):
        assert_close(orc.SUFFSTATS[cov](g['X']), g[f'stats.{cov}'], 1e-15)


# --- G1/G2/G3/G11: GMM VB iterations -----------------------------------------

GMM_CASES = [('g01_gmm_diag_c1', TOL64), ('g02_gmm_full', 1e-9),
             ('g03_gmm_iso', TOL64), ('g11_gmm_diag_c1_f32', TOL32),
             ('g11_gmm_full_f32', 2e-3)]


@pytest.mark.parametrize('name,tol', GMM_CASES)
def test_gmm_iterations(name, tol):
    g = load_golden(name)
    cov = str(g['cov_type'])
    X = g['X']
    f = orc.FAMILIES[cov]
    post, prior = std_params(g, 'init.p0.posterior'), std_params(g, 'init.p0.prior')
    (w_post,), (w_prior,) = std_params(g, 'init.p1.posterior'), std_params(g, 'init.p1.prior')
    D = X.shape[1]

    # detailed first-step intermediates
    stats = orc.SUFFSTATS[cov](X)
    exp_T = f['exp'](*post)
    assert_close(exp_T, g['exp_T'], tol, 'E[T]')
    assert_close(orc.normal_llh(stats, exp_T, D), g['pc_llh'], tol, 'pc_llh')
    lw = orc.log_weights(w_post)
    assert_close(lw, g['log_weights'].reshape(-1), tol, 'log_weights')
    per_frame, resps = orc.mixture_estep(stats, exp_T, D, lw)
    assert_close(per_frame, g['per_frame'], tol, 'per_frame')
    assert_close(resps, g['resps'], tol * 10, 'resps')
    assert_close(f['nat'](*post), g['nat_post'], tol, 'nat_post')
    assert_close(f['lnorm'](*post), g['lognorm_post'], tol, 'lognorm')

    for it in range(int(g['niter'])):
        r = orc.gmm_elbo_step(X, cov, post, prior, w_post, w_prior)
        if it == 0:
            assert_close(r['kl'], g['kl'], max(tol, 1e-9) * 10, 'kl')
            assert_close(r['acc_normal'], g['acc0.p0'], tol, 'acc normal')
            assert_close(r['acc_weights'], g['acc0.p1'], tol, 'acc weights')
        assert_close(r['value'], g['elbos'][it], tol, f'elbo it{it}')
        post, w_post = orc.gmm_mstep(cov, post, prior, w_post, w_prior,
                                     r['acc_normal'], r['acc_weights'])
        if X.dtype == np.float32 and it > 0:
            continue        # fp32 trajectories diverge; per-step parity only
        for arr, ref in zip(post, std_params(g, f'it{it}.p0.posterior')):
            assert_close(arr.reshape(ref.shape), ref, tol * 100, f'post it{it}')
        assert_close(w_post, g[f'it{it}.p1.posterior.concentrations'], tol * 10)
        if X.dtype == np.float32:
            # restart from the reference's posterior: per-step parity
            post = std_params(g, f'it{it}.p0.posterior')
            w_post = g[f'it{it}.p1.posterior.concentrations']


def test_gmm_labels_branch():
    g = load_golden('g01_gmm_labels')
    post, prior = std_params(g, 'init.p0.posterior'), std_params(g, 'init.p0.prior')
    (w_post,), (w_prior,) = std_params(g, 'init.p1.posterior'), std_params(g, 'init.p1.prior')
    r = orc.gmm_elbo_step(g['X'], 'full', post, prior, w_post, w_prior,
                          labels=g['labels'])
    assert_close(r['value'], g['elbo'], TOL64)
    assert_close(r['acc_normal'], g['acc0.p0'], TOL64)
    assert_close(r['acc_weights'], g['acc0.p1'], TOL64)


# --- G4/G7: HMM ---------------------------------------------------------------

def _graph(g, prefix):
    return dict(init=g[prefix + '.init'], final=g[prefix + '.final'],
                trans=g[prefix + '.trans'].copy(),
                order=g[prefix + '.pdf_id_mapping'])


def _normal_group(g, prefix, i=0):
    cov = COV_OF[dist_cls(g, f'{prefix}.p{i}.posterior')]
    post = std_params(g, f'{prefix}.p{i}.posterior')
    return dict(cov_type=cov, post=post, prior=std_params(g, f'init.p{i}.prior'),
                S=len(post[0]), G=0)


@pytest.mark.parametrize('cov', ['full', 'diagonal', 'isotropic'])
@pytest.mark.parametrize('suffix,tol', [('', 1e-9), ('_f32', 2e-3)])
def test_g4_hmm(cov, suffix, tol):
    g = load_golden(f'g04_hmm_{cov}{suffix}')
    X, graph = g['X'], _graph(g, 'graph')
    groups = [_normal_group(g, 'init')]
    pc_all, _ = orc.emissions_estep(X, groups)
    assert_close(pc_all, g['pc_llhs'], tol, 'pc_llhs')
    pc = g['pc_llhs']
    assert_close(orc.forward(pc, graph['init'], graph['trans']), g['log_alphas'], tol)
    assert_close(orc.backward(pc, graph['final'], graph['trans']), g['log_betas'], tol)
    gamma, xi, lnm = orc.posteriors(pc, graph['init'], graph['final'], graph['trans'], True)
    assert_close(gamma, g['gamma'], tol * 10, 'gamma')
    assert_close(xi.sum(0), g['xi_sum'], tol * 10, 'xi_sum')
    assert_close(xi[:3], g['xi_first'], tol * 10, 'xi_first')
    assert_close(lnm, g['lognorm_mean'], tol)
    np.testing.assert_array_equal(
        orc.best_path(pc, graph['init'], graph['final'], graph['trans']),
        orc.best_path(pc, graph['init'], graph['final'], graph['trans']))
    for it in range(3):
        r = orc.hmm_elbo_step(X, groups, graph, datasize=len(X), trans_posteriors=True)
        assert_close(r['value'], g['elbos'][it], tol * 10, f'elbo {it}')
        if it == 0:
            assert_close(r['acc'][0][0], g['acc0.p0'], tol * 10, 'acc')
        groups = orc.emissions_mstep(groups, r['acc'], 1.)
        if suffix:
            groups = [_normal_group(g, f'it{it}')]
            continue
        for arr, ref in zip(groups[0]['post'], std_params(g, f'it{it}.p0.posterior')):
            assert_close(arr.reshape(ref.shape), ref, 1e-7, f'post it{it}')
    if not suffix:
        # decode after 3 iterations (bit-exact target: int64 path)
        pc_all, _ = orc.emissions_estep(X, groups)
        path = orc.best_path(pc_all[:, graph['order']], graph['init'],
                             graph['final'], graph['trans'])
        np.testing.assert_array_equal(graph['order'][path], g['decode'])
        post, _ = orc.posteriors(pc_all[:, graph['order']], graph['init'],
                                 graph['final'], graph['trans'])
        assert_close(post, g['posteriors'], 1e-6)


# --- G18: the prior of a VAE with one sample per frame --------------------------

@pytest.mark.parametrize('cov', ['full', 'diagonal', 'isotropic'])
@pytest.mark.parametrize('kind', ['gmm', 'hmm'])
def test_g18_prior_value_and_gradient_wrt_samples(kind, cov):
    """The reference's value, autograd gradient w.r.t. the samples and accumulated statistics of
    a GMM / HMM prior over phi(z_t) (vae.py:63-86 with nsamples = 1)."""
    g = load_golden(f'g18_onesample_{kind}_{cov}')
    Z, c = g['z'][:, 0, :], g['c']
    post = std_params(g, 'init.p0.posterior')
    if kind == 'gmm':
        (w_post,) = std_params(g, 'init.p1.posterior')
        value, weights, exp_T = orc.vae_gmm_prior(cov, Z, post, w_post)
    else:
        value, weights, exp_T = orc.vae_hmm_prior(cov, Z, post, _graph(g, 'graph'))
    assert_close(value, g['exp_llh'].reshape(-1), TOL64, 'exp_llh')
    grad = orc.prior_gradient_wrt_samples(cov, Z, weights, exp_T, c)
    assert_close(grad, g['grad_z'][:, 0, :], 1e-9, 'd/dz')
    acc = weights.T @ orc.SUFFSTATS[cov](Z)
    assert_close(acc, g['acc.p0'].reshape(acc.shape), 1e-9, 'acc')


def test_g7_viterbi_ties():
    g = load_golden('g07_viterbi_ties')
    graph = _graph(g, 'graph')
    for i in range(3):
        l = g[f'llhs{i}']
        np.testing.assert_array_equal(
            orc.best_path(l, graph['init'], graph['final'], graph['trans']), g[f'path{i}'])
        gamma, xi, _ = orc.posteriors(l, graph['init'], graph['final'], graph['trans'], True)
        assert_close(gamma, g[f'gamma{i}'], 1e-12)
        assert_close(xi.sum(0), g[f'xi_sum{i}'], 1e-12)


@pytest.mark.parametrize('branch', ['viterbi', 'state_path'])
def test_g7_hmm_hard_alignment(branch):
    g = load_golden(f'g07_hmm_{branch}')
    groups = [_normal_group(g, 'init')]
    kw = {'viterbi': True} if branch == 'viterbi' else {'state_path': g['state_path']}
    r = orc.hmm_elbo_step(g['X'], groups, _graph(g, 'graph'), datasize=len(g['X']),
                          trans_posteriors=True, **kw)
    assert_close(r['value'], g['elbo'], 1e-10)
    assert_close(r['acc'][0][0], g['acc0.p0'], 1e-10)


# --- G5/G6/G8: PhoneLoop ------------------------------------------------------

def _ploop_groups(g, prefix):
    groups, i = [], 0
    for S, G in zip(g['group_sizes'], g['group_ncomp']):
        cov = COV_OF[dist_cls(g, f'{prefix}.p{i}.posterior')]
        groups.append(dict(
            cov_type=cov, S=int(S), G=int(G),
            post=std_params(g, f'{prefix}.p{i}.posterior'),
            prior=std_params(g, f'init.p{i}.prior'),
            w_post=g[f'{prefix}.p{i + 1}.posterior.concentrations'],
            w_prior=g[f'init.p{i + 1}.prior.concentrations']))
        i += 2
    return groups, i


@pytest.mark.parametrize('kind', ['dirichlet', 'dirichlet_process',
                                  'gamma_dirichlet_process'])
def test_g5_phoneloop(kind):
    g = load_golden(f'g05_phoneloop_{kind}')
    X, graph = g['X'], _graph(g, 'graph')
    groups, ci = _ploop_groups(g, 'init')
    start, end = g['start_idxs'], g['end_idxs']
    P = len(start)
    state = dict(post=g[f'init.p{ci}.posterior.concentrations'],
                 prior=g[f'init.p{ci}.prior.concentrations'].copy(),
                 ordering=np.arange(P))
    if kind == 'gamma_dirichlet_process':
        state.update(g_prior_shape=g['init.concentration.prior.shape'],
                     g_prior_rate=g['init.concentration.prior.rate'],
                     g_post_shape=g['init.concentration.posterior.shape'],
                     g_post_rate=g['init.concentration.posterior.rate'])
    for it in range(2):
        r = orc.hmm_elbo_step(X, groups, graph, datasize=len(X), trans_posteriors=True,
                              extra_kl=orc.categorical_kl(kind, state))
        if it == 0:
            assert_close(r['exp_llh'], g['exp_llh'], 1e-9, 'exp_llh')
            assert_close(r['resps'], g['gamma'], 1e-8, 'gamma')
            assert_close(r['trans_resps'].sum(0), g['xi_sum'], 1e-8, 'xi_sum')
        assert_close(r['value'], g['elbos'][it], 1e-9, f'elbo {it}')
        counts = orc.phone_counts(r['trans_resps'], r['resps'], start, end)
        cstats = orc.cat_suffstats(counts.reshape(1, -1)).sum(0) if kind == 'dirichlet' \
            else counts
        assert_close(cstats, g[f'acc{it}.p{ci}'], 1e-8, 'phone stats')
        for k, (ns, ws) in enumerate(r['acc']):
            assert_close(ns, g[f'acc{it}.p{2 * k}'], 1e-8, f'acc normal {k}')
            assert_close(ws, g[f'acc{it}.p{2 * k + 1}'], 1e-8, f'acc weights {k}')
        groups = orc.emissions_mstep(groups, r['acc'], 1.)
        state = orc.categorical_mstep(kind, state, cstats)
        lw = orc.categorical_log_weights(kind, state)
        graph['trans'] = orc.phoneloop_update_trans(graph['trans'], lw, start, end)
        assert_close(state['post'], g[f'it{it}.p{ci}.posterior.concentrations'], 1e-9)
        assert_close(np.exp(graph['trans']), np.exp(g[f'it{it}.trans']), 1e-9, 'trans')
        if kind != 'dirichlet':
            np.testing.assert_array_equal(state['ordering'], g[f'it{it}.ordering'])
        if kind == 'gamma_dirichlet_process':
            assert_close(state['g_post_rate'], g[f'it{it}.concentration.posterior.rate'], 1e-10)
            assert_close(state['g_post_shape'], g[f'it{it}.concentration.posterior.shape'], 1e-10)
        for k, grp in enumerate(groups):
            for arr, ref in zip(grp['post'], std_params(g, f'it{it}.p{2 * k}.posterior')):
                assert_close(arr.reshape(ref.shape), ref, 1e-8)
    pc_all, _ = orc.emissions_estep(X, groups)
    path = orc.best_path(pc_all[:, graph['order']], graph['init'], graph['final'],
                         graph['trans'])
    np.testing.assert_array_equal(graph['order'][path], g['decode'])


def test_g6_alignment_graph_and_g8_joint():
    g = load_golden('g06_phoneloop_ali')
    X = g['X']
    groups, ci = _ploop_groups(g, 'init')
    pc_all, _ = orc.emissions_estep(X, groups)
    assert_close(pc_all, g['joint_pc_llh'], 1e-10, 'G8 joint pc llh')
    ali = _graph(g, 'ali')
    assert len(set(ali['order'].tolist())) < len(ali['order'])   # repeated pdf ids
    scale = float(g['scale'])
    w_post = g[f'init.p{ci}.posterior.concentrations']
    w_prior = g[f'init.p{ci}.prior.concentrations']
    r = orc.hmm_elbo_step(X, groups, ali, datasize=1000, scale=scale,
                          extra_kl=orc.dir_kl(w_post, w_prior).sum())
    assert_close(r['exp_llh'], g['exp_llh'], 1e-9)
    assert_close(r['resps'], g['gamma'], 1e-8)
    assert_close(r['value'], g['elbo'], 1e-9)
    for k, (ns, ws) in enumerate(r['acc']):
        assert_close(ns, g[f'acc0.p{2 * k}'], 1e-8)
        assert_close(ws, g[f'acc0.p{2 * k + 1}'], 1e-8)
    assert np.all(g[f'acc0.p{ci}'] == 0)          # phoneloop.py:98-100
    # decode / posteriors were taken after the update (make_golden.py g6_g8)
    groups = orc.emissions_mstep(groups, r['acc'], r['value'] * 0 + 1000. / len(X))
    for k, grp in enumerate(groups):
        for arr, ref in zip(grp['post'], std_params(g, f'it0.p{2 * k}.posterior')):
            assert_close(arr.reshape(ref.shape), ref, 1e-8)
    pc_all, _ = orc.emissions_estep(X, groups)
    pc = pc_all.dtype.type(scale) * pc_all[:, ali['order']]
    path = orc.best_path(pc, ali['init'], ali['final'], ali['trans'])
    np.testing.assert_array_equal(ali['order'][path], g['decode_ali'])
    pc_all, _ = orc.emissions_estep(X, groups, stats_scale=scale)    # Q5
    post, _ = orc.posteriors(pc_all[:, ali['order']], ali['init'], ali['final'],
                             ali['trans'])
    assert_close(post, g['posteriors_ali'], 1e-8)


# --- G9: ELBO bookkeeping -------------------------------------------------------

def test_g9_bookkeeping():
    g = load_golden('g09_elbo_bookkeeping')
    X, lens, N = g['X'], g['lens'], int(g['datasize'])
    post, prior = std_params(g, 'init.p0.posterior'), std_params(g, 'init.p0.prior')
    (w_post,), (w_prior,) = std_params(g, 'init.p1.posterior'), std_params(g, 'init.p1.prior')
    off = np.concatenate([[0], np.cumsum(lens)])
    total, acc_n, acc_w = 0., 0., 0.
    for u in range(len(lens)):
        r = orc.gmm_elbo_step(X[off[u]:off[u + 1]], 'diagonal', post, prior,
                              w_post, w_prior, datasize=N)
        assert_close(r['value'], g['utt_values'][u], 1e-10)
        total += r['value']
        acc_n, acc_w = acc_n + r['acc_normal'], acc_w + r['acc_weights']
    assert_close(total, g['sum_value'], 1e-10)                        # Q1: -U*KL
    assert_close(total / (len(lens) * N), g['logged'], 1e-10)
    assert_close(acc_n, g['acc_sum.p0'], 1e-10)
    scale = N / float(lens.sum())                                     # Q2
    assert_close(scale * acc_n, g['stored.p0'], 1e-10)
    assert_close(scale * acc_w, g['stored.p1'], 1e-10)
    post1, w1 = orc.gmm_mstep('diagonal', post, prior, w_post, w_prior,
                              scale * acc_n, scale * acc_w)
    for arr, ref in zip(post1, std_params(g, 'it0.p0.posterior')):
        assert_close(arr.reshape(ref.shape), ref, 1e-9)
    r = orc.gmm_elbo_step(X, 'diagonal', post1, prior, w1, w_prior, datasize=N)
    post2, w2 = orc.gmm_mstep('diagonal', post1, prior, w1, w_prior,
                              r['scale'] * r['acc_normal'],
                              r['scale'] * r['acc_weights'], lrate=.3)
    for arr, ref in zip(post2, std_params(g, 'it1_lr03.p0.posterior')):
        assert_close(arr.reshape(ref.shape), ref, 1e-9)
    assert_close(w2, g['it1_lr03.p1.posterior.concentrations'], 1e-9)


# --- the torch-CPU timing port agrees with the numpy oracle ------------------------

def test_torch_port_matches_oracle():
    import torch
    from oracle import torch_port as tp
    g = load_golden('g02_gmm_full')
    X = g['X']
    post, prior = std_params(g, 'init.p0.posterior'), std_params(g, 'init.p0.prior')
    (w_post,), (w_prior,) = std_params(g, 'init.p1.posterior'), std_params(g, 'init.p1.prior')
    tt = lambda arrs: tuple(torch.from_numpy(a.copy()) for a in arrs)
    value, new_post, new_w = tp.gmm_iteration(torch.from_numpy(X), tt(post), tt(prior),
                                              torch.from_numpy(w_post.copy()),
                                              torch.from_numpy(w_prior.copy()), chunk=len(X))
    assert_close(value, g['elbos'][0], 1e-10, 'torch port elbo vs reference')
    for arr, ref in zip(new_post, std_params(g, 'it0.p0.posterior')):
        assert_close(arr.numpy().reshape(ref.shape), ref, 1e-8)
    assert_close(new_w.numpy(), g['it0.p1.posterior.concentrations'], 1e-10)


def test_torch_port_diag_gmm_matches_the_reference_golden():
    """bench.py's config-1 cpu_baseline (oracle/torch_port.py: gmm_diag_iteration) over the five
    iterations of the reference's own run of BASELINE config 1 (G1: K = 8, D = 2, T = 1000)."""
    import torch
    from oracle import torch_port as tp
    g = load_golden('g01_gmm_diag_c1')
    X = torch.from_numpy(g['X'])
    tt = lambda arrs: tuple(torch.from_numpy(a.copy()) for a in arrs)       # noqa: E731
    post, prior = tt(std_params(g, 'init.p0.posterior')), tt(std_params(g, 'init.p0.prior'))
    w_post = torch.from_numpy(std_params(g, 'init.p1.posterior')[0].copy())
    w_prior = torch.from_numpy(std_params(g, 'init.p1.prior')[0].copy())
    for it in range(int(g['niter'])):
        value, post, w_post = tp.gmm_diag_iteration(X, post, prior, w_post, w_prior)
        assert_close(value, g['elbos'][it], 1e-10, f'elbo {it}')
        for arr, ref in zip(post, std_params(g, f'it{it}.p0.posterior')):
            assert_close(arr.numpy().reshape(ref.shape), ref, 1e-8, f'posterior {it}')
        assert_close(w_post.numpy(), g[f'it{it}.p1.posterior.concentrations'], 1e-10)


def test_torch_port_hmm_matches_oracle():
    '''bench.py's config-3 cpu_baseline (oracle/torch_port.py: hmm_elbo, the
    reference's op sequence on torch CPU tensors) against the numpy oracle, which
    is pinned on the reference's goldens G4-G6.'''
    import torch
    from oracle import torch_port as tp
    rng = np.random.RandomState(0)
    P, G, D, T = 5, 4, 6, 40
    S, K = 3 * P, 3 * P * G
    post = (rng.randn(K, D), 1 + rng.rand(K, 1), 2 + rng.rand(K, 1), 1 + rng.rand(K, D))
    prior = (rng.randn(K, D), np.ones((K, 1)), np.ones((K, 1)), np.ones((K, D)))
    wp, w0 = 1 + rng.rand(S, G), np.ones((S, G))
    trans = np.full((S, S), -np.inf)
    for s in range(S):
        trans[s, s] = np.log(.75)
        if s % 3 < 2:
            trans[s, s + 1] = np.log(.25)
        else:
            trans[s, ::3] = np.log(.25 / P)
    init = np.where(np.arange(S) % 3 == 0, np.log(1. / P), -np.inf)
    fin = np.where(np.arange(S) % 3 == 2, np.log(.25), -np.inf)
    X = rng.randn(T, D)
    groups = [dict(cov_type='diagonal', S=S, G=G, post=post, prior=prior, w_post=wp, w_prior=w0)]
    ref = orc.hmm_elbo_step(X, groups, dict(init=init, final=fin, trans=trans, order=np.arange(S)),
                            datasize=1000, trans_posteriors=True)
    t = lambda a: torch.from_numpy(np.asarray(a))        # noqa: E731
    value, acc, wstats, xi = tp.hmm_elbo(t(X), tuple(map(t, post)), tuple(map(t, prior)), t(wp),
                                         t(w0), t(init), t(fin), t(trans), 1000)
    assert_close(float(value), ref['value'], 1e-12, 'value')
    assert_close(acc.numpy(), ref['acc'][0][0], 1e-12, 'Gaussian statistics')
    assert_close(wstats.numpy(), ref['acc'][0][1], 1e-12, 'weight statistics')
    # ... and through an alignment graph whose pdf ids repeat (phones 1, 0, 1)
    order = np.asarray([3, 4, 5, 0, 1, 2, 3, 4, 5])
    Sa = len(order)
    ta = np.full((Sa, Sa), -np.inf)
    for s_ in range(Sa):
        ta[s_, s_] = np.log(.75)
        if s_ + 1 < Sa:
            ta[s_, s_ + 1] = np.log(.25)
    ia = np.where(np.arange(Sa) == 0, 0., -np.inf)
    fa = np.where(np.arange(Sa) == Sa - 1, np.log(.25), -np.inf)
    ref_a = orc.hmm_elbo_step(X, groups, dict(init=ia, final=fa, trans=ta, order=order), datasize=1000)
    value, acc, wstats, _ = tp.hmm_elbo(t(X), tuple(map(t, post)), tuple(map(t, prior)), t(wp), t(w0),
                                        t(ia), t(fa), t(ta), 1000, trans_posteriors=False,
                                        order=order.tolist())
    assert_close(float(value), ref_a['value'], 1e-12, 'value (alignment graph)')
    assert_close(acc.numpy(), ref_a['acc'][0][0], 1e-12, 'Gaussian statistics (alignment graph)')
    assert_close(wstats.numpy(), ref_a['acc'][0][1], 1e-12, 'weight statistics (alignment graph)')
    assert_close(xi.numpy(), ref['trans_resps'].sum(0), 1e-12, 'transition posteriors')


def test_torch_port_hmm_with_two_groups_matches_oracle():
    """bench.py's config-5 cpu_baseline (oracle/torch_port.py: hmm_elbo_groups -- emissions that
    are a JointModelSet of two MixtureSets, as the recipes build them: mkphones.py:100-113) against
    the numpy oracle's hmm_elbo_step with the same two groups, through an alignment-style graph
    whose pdf ids reach into both groups and repeat."""
    import torch
    from oracle import torch_port as tp
    rng = np.random.RandomState(1)
    D, T = 5, 50
    shapes = [(2, 5), (6, 3)]                     # (states, Gaussians per state) of the two groups
    groups, tgroups = [], []
    t = lambda a: torch.from_numpy(np.asarray(a))        # noqa: E731
    for S, G in shapes:
        K = S * G
        post = (rng.randn(K, D), 1 + rng.rand(K, 1), 2 + rng.rand(K, 1), 1 + rng.rand(K, D))
        prior = (rng.randn(K, D), np.ones((K, 1)), np.ones((K, 1)), np.ones((K, D)))
        wp, w0 = 1 + rng.rand(S, G), np.ones((S, G))
        groups.append(dict(cov_type='diagonal', S=S, G=G, post=post, prior=prior, w_post=wp, w_prior=w0))
        tgroups.append((tuple(map(t, post)), tuple(map(t, prior)), t(wp), t(w0)))
    order = np.asarray([0, 1, 5, 6, 7, 2, 3, 4, 5, 6, 7, 0, 1])
    Sa = len(order)
    ta = np.full((Sa, Sa), -np.inf)
    for s_ in range(Sa):
        ta[s_, s_] = np.log(.75)
        if s_ + 1 < Sa:
            ta[s_, s_ + 1] = np.log(.25)
    ia = np.where(np.arange(Sa) == 0, 0., -np.inf)
    fa = np.where(np.arange(Sa) == Sa - 1, np.log(.25), -np.inf)
    X = rng.randn(T, D)
    ref = orc.hmm_elbo_step(X, groups, dict(init=ia, final=fa, trans=ta, order=order), datasize=777)
    value, accs, _ = tp.hmm_elbo_groups(t(X), tgroups, t(ia), t(fa), t(ta), 777,
                                        trans_posteriors=False, order=order.tolist())
    assert_close(float(value), ref['value'], 1e-12, 'value')
    for i, (acc, wstats) in enumerate(accs):
        assert_close(acc.numpy(), ref['acc'][i][0], 1e-12, f'Gaussian statistics, group {i}')
        assert_close(wstats.numpy(), ref['acc'][i][1], 1e-12, f'weight statistics, group {i}')


@pytest.mark.parametrize('cov', ['diagonal', 'full'])
def test_torch_port_vae_prior_matches_oracle(cov):
    """bench.py's config-4 cpu_baseline (oracle/torch_port.py: vae_hmm_prior_path -- the
    reference's op sequence incl. torch autograd) against the numpy oracle's `vae_hmm_prior`
    and `prior_gradient_wrt_samples`, which the CPU suite pins on the reference's G18 goldens."""
    import torch
    from oracle import torch_port as tp
    rng = np.random.RandomState(3)
    S, D, T = 6, 5, 37
    if cov == 'full':
        A = rng.randn(S, D, D) * .3
        W = np.einsum('sij,skj->sik', A, A) + np.eye(D) * .5
        post = (rng.randn(S, D), 1 + rng.rand(S, 1), W, D + 1 + rng.rand(S, 1))
    else:
        post = (rng.randn(S, D), 1 + rng.rand(S, 1), 2 + rng.rand(S, 1), 1 + rng.rand(S, D))
    trans = np.full((S, S), -np.inf)
    for s_ in range(S):
        trans[s_, s_] = np.log(.6)
        trans[s_, (s_ + 1) % S] = np.log(.4)
    init = np.where(np.arange(S) == 0, 0., -np.inf)
    fin = np.where(np.arange(S) >= S - 2, np.log(.5), -np.inf)
    Z = rng.randn(T, D) * 1.3
    value, resps, exp_T = orc.vae_hmm_prior(cov, Z, post, dict(init=init, final=fin, trans=trans,
                                                                order=np.arange(S)))
    grad = orc.prior_gradient_wrt_samples(cov, Z, resps, exp_T)
    t = lambda a: torch.from_numpy(np.asarray(a))          # noqa: E731
    v, g, acc = tp.vae_hmm_prior_path(t(Z), cov, tuple(map(t, post)), t(init), t(fin), t(trans))
    assert_close(v.numpy(), value, 1e-12, 'per-frame value')
    assert_close(g.numpy(), grad, 1e-11, 'gradient w.r.t. the samples')
    assert_close(acc.numpy(), resps.T @ orc.SUFFSTATS[cov](Z), 1e-12, 'accumulated statistics')
