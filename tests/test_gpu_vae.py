"""GPU parity of the "statistics-in" path (VAE models, SURVEY.md section 8
row f.2): dense [T, Q] statistics into the prior, gradient back to the
statistics / frames, accumulation from dense statistics.  Goldens G14 were
produced by the reference (tests/golden/make_golden.py:g14_vae)."""

import numpy as np
import pytest
import torch

from helpers import assert_close, load_golden

pytestmark = pytest.mark.gpu

TOL = 1e-9          # fp64 kernels against the fp64 reference


def _prior(g, kind):
    import beer_amd as beer
    from gpu_helpers import build_hmm, build_mixture, build_param
    if kind == 'gmm':
        return build_mixture(g)
    if kind == 'hmm':
        return build_hmm(g)
    return beer.Normal(build_param(g, 'init.p0'))


@pytest.mark.parametrize('name,kind', [
    ('g14_statsin_gmm_full', 'gmm'), ('g14_statsin_gmm_diagonal', 'gmm'),
    ('g14_statsin_gmm_isotropic', 'gmm'), ('g14_statsin_hmm_full', 'hmm'),
    ('g14_statsin_hmm_diagonal', 'hmm'), ('g14_statsin_hmm_isotropic', 'hmm'),
    ('g14_statsin_normal_full', 'normal')])
def test_statsin_prior(name, kind):
    from gpu_helpers import npy, params_of, tt
    g = load_golden(name)
    prior = _prior(g, kind)
    z = tt(g['z']).requires_grad_(True)
    T, ns, Dz = z.shape
    c = tt(g['c'])
    flat = prior.sufficient_statistics(z.view(-1, Dz))
    assert isinstance(flat, torch.Tensor) and flat.requires_grad
    stats = flat.reshape(T, ns, -1).mean(dim=1)
    stats.retain_grad()
    assert_close(npy(stats), g['stats'], 1e-12, 'stats')
    exp_llh = prior.expected_log_likelihood(stats)
    assert_close(npy(exp_llh).reshape(g['exp_llh'].shape), g['exp_llh'], TOL, 'exp_llh')
    # same expression as the generator (for Normal the [T,1] value broadcasts)
    (c * exp_llh.reshape(g['exp_llh'].shape)).sum().backward()
    assert_close(npy(stats.grad), g['grad_stats'], TOL, 'd/dstats')
    assert_close(npy(z.grad), g['grad_z'], TOL, 'd/dz')
    acc = prior.accumulate(stats.detach())
    for i, p in enumerate(params_of(prior)):
        assert_close(npy(acc[p]).reshape(g[f'acc.p{i}'].shape), g[f'acc.p{i}'], TOL, f'acc p{i}')


def test_statsin_matches_frames_path():
    'Dense statistics of plain frames give the same E-step as the fused path.'
    import beer_amd as beer
    from gpu_helpers import npy
    torch.manual_seed(0)
    X = torch.randn(3000, 7, dtype=torch.float64, device='cuda')
    for cov in ('full', 'diagonal', 'isotropic'):
        ns = beer.NormalSet.create(torch.zeros(7, dtype=torch.float64),
                                   torch.ones(7, dtype=torch.float64), size=12, cov_type=cov)
        for model in (beer.Mixture.create(ns), beer.MixtureSet.create(3, ns)):
            lazy = model.sufficient_statistics(X)
            a = model.expected_log_likelihood(lazy)
            if isinstance(model, beer.MixtureSet):
                sr = torch.rand(len(X), 3, dtype=torch.float64, device='cuda')
                acc_a = model.accumulate(lazy, sr)
            else:
                acc_a = model.accumulate(lazy)
            model.clear_cache()
            dense = lazy.dense()
            b = model.expected_log_likelihood(dense)
            acc_b = model.accumulate(dense, sr) if isinstance(model, beer.MixtureSet) \
                else model.accumulate(dense)
            model.clear_cache()
            assert_close(npy(b), npy(a), 1e-10, f'{cov} llh')
            for p in acc_a:
                assert_close(npy(acc_b[p]), npy(acc_a[p]), 1e-9, f'{cov} acc')


def test_vae_step_against_reference(monkeypatch):
    'One ELBO + backward of a GMM-VAE with the reference\'s weights and noise.'
    import beer_amd as beer
    from beer_amd.dists import normaldiag
    from gpu_helpers import build_mixture, npy, params_of, tt
    g = load_golden('g14_vae_gmm_step')
    X = tt(g['X'])
    Dx, Dz = X.shape[1], g['noise'].shape[-1]
    enc = beer.nnet.ResidualFeedForwardNet(dim_in=Dx, nblocks=2, block_width=8)
    dec = beer.nnet.ResidualFeedForwardNet(dim_in=Dz, nblocks=2, block_width=8)
    vae = beer.VAE(build_mixture(g), enc, dec, reference_broadcast=True).double().to('cuda')
    with torch.no_grad():
        for name, p in vae.named_parameters():
            p.copy_(tt(g['nn.' + name]))
    monkeypatch.setattr(normaldiag, '_randn', lambda *a, **k: tt(g['noise']))
    elbo = beer.evidence_lower_bound(vae, X, nsamples=int(g['nsamples']),
                                     datasize=int(g['datasize']))
    assert abs(float(elbo) - float(g['elbo'])) <= 1e-9 * abs(float(g['elbo']))
    elbo.backward()
    for name, p in vae.named_parameters():
        assert_close(npy(p.grad), g['nngrad.' + name], 1e-7, 'grad ' + name)
    for i, p in enumerate(params_of(vae)):
        assert_close(npy(elbo._acc_stats[p]).reshape(g[f'acc.p{i}'].shape), g[f'acc.p{i}'],
                     1e-9, f'acc p{i}')


def test_hmm_vae_step_at_config4_dimensions(monkeypatch):
    '''BASELINE config 4's model at its own dimensions -- D = 40, 64-dimensional latent
    variable, HMM prior with diagonal Gaussians -- one ELBO + backward with the reference's
    weights and noise: value, gradients of every network weight, accumulated statistics
    (golden g14_hmm_vae_step; vae.py:63-89, hmm.py:73-100).'''
    import beer_amd as beer
    from beer_amd.dists import normaldiag
    from gpu_helpers import build_hmm, npy, params_of, tt
    g = load_golden('g14_hmm_vae_step')
    X = tt(g['X'])
    Dx, Dz = X.shape[1], g['noise'].shape[-1]
    assert (Dx, Dz) == (40, 64)
    enc = beer.nnet.ResidualFeedForwardNet(dim_in=Dx, nblocks=2, block_width=32)
    dec = beer.nnet.ResidualFeedForwardNet(dim_in=Dz, nblocks=2, block_width=32)
    vae = beer.VAE(build_hmm(g), enc, dec, reference_broadcast=True).double().to('cuda')
    with torch.no_grad():
        for name, p in vae.named_parameters():
            p.copy_(tt(g['nn.' + name]))
    monkeypatch.setattr(normaldiag, '_randn', lambda *a, **k: tt(g['noise']))
    elbo = beer.evidence_lower_bound(vae, X, nsamples=int(g['nsamples']),
                                     datasize=int(g['datasize']))
    assert abs(float(elbo) - float(g['elbo'])) <= 1e-9 * abs(float(g['elbo']))
    elbo.backward()
    for name, p in vae.named_parameters():
        assert_close(npy(p.grad), g['nngrad.' + name], 1e-7, 'grad ' + name)
    for i, p in enumerate(params_of(vae)):
        assert_close(npy(elbo._acc_stats[p]).reshape(g[f'acc.p{i}'].shape), g[f'acc.p{i}'],
                     1e-9, f'acc p{i}')


def test_vae_trains():
    'HMM-VAE (config 4 shape, reduced): the ELBO improves over a few epochs.'
    import beer_amd as beer
    torch.manual_seed(1)
    T, Dx, Dz = 400, 6, 3
    X = torch.randn(T, Dx, device='cuda') * 2. + 1.
    graph = beer.graph.Graph()
    s0, s4 = graph.add_state(), graph.add_state()
    graph.start_state, graph.end_state = s0, s4
    states = [graph.add_state(pdf_id=i) for i in range(3)]
    graph.add_arc(s0, states[0])
    for i, s in enumerate(states):
        graph.add_arc(s, s)
        graph.add_arc(s, states[(i + 1) % 3])
    graph.add_arc(states[2], s4)
    graph.normalize()
    ns = beer.NormalSet.create(torch.zeros(Dz), torch.ones(Dz), size=3, cov_type='full')
    vae = beer.VAE(beer.HMM.create(graph.compile(), ns),
                   beer.nnet.ResidualFeedForwardNet(Dx, 2, 16),
                   beer.nnet.ResidualFeedForwardNet(Dz, 2, 16)).to('cuda')
    cjg = beer.VBConjugateOptimizer(vae.mean_field_factorization(), lrate=.1)
    std = torch.optim.Adam(vae.parameters(), lr=1e-2)
    optim = beer.VBOptimizer(cjg, std)
    values = []
    for _ in range(30):
        optim.init_step()
        elbo = beer.evidence_lower_bound(vae, X, nsamples=5)
        elbo.backward()
        optim.step()
        values.append(float(elbo))
    assert np.isfinite(values).all()
    assert np.mean(values[-5:]) > np.mean(values[:5])


def test_vae_batch_equals_per_utterance_loop(monkeypatch):
    '''accumulate_elbo on a minibatch of utterances == the reference's loop of
    per-utterance evidence_lower_bound calls (value, gradients, statistics).'''
    import beer_amd as beer
    from beer_amd.dists import normaldiag
    from gpu_helpers import npy
    torch.manual_seed(5)
    Dx, Dz, nsamp = 5, 3, 2
    lengths = [30, 17, 44]
    X = torch.randn(sum(lengths), Dx, dtype=torch.float64, device='cuda')
    master = torch.randn(sum(lengths), nsamp, Dz, dtype=torch.float64, device='cuda')
    graph = beer.graph.Graph()
    s0, s4 = graph.add_state(), graph.add_state()
    graph.start_state, graph.end_state = s0, s4
    st = [graph.add_state(pdf_id=i) for i in range(3)]
    graph.add_arc(s0, st[0])
    for i, s in enumerate(st):
        graph.add_arc(s, s)
        graph.add_arc(s, st[(i + 1) % 3])
    graph.add_arc(st[2], s4)
    graph.normalize()
    ns = beer.NormalSet.create(torch.zeros(Dz, dtype=torch.float64),
                               torch.ones(Dz, dtype=torch.float64), size=3,
                               cov_type='diagonal')
    vae = beer.VAE(beer.HMM.create(graph.compile(), ns),
                   beer.nnet.ResidualFeedForwardNet(Dx, 1, 8),
                   beer.nnet.ResidualFeedForwardNet(Dz, 1, 8)).double().to('cuda')
    cursor = [0]

    def fake_randn(n, *rest, **conf):
        out = master[cursor[0]:cursor[0] + n]
        cursor[0] += n
        return out
    monkeypatch.setattr(normaldiag, '_randn', fake_randn)
    N = 1000
    loop = beer.evidence_lower_bound(datasize=N)
    for u in torch.split(X, lengths):
        loop += beer.evidence_lower_bound(vae, u, datasize=N, nsamples=nsamp)
    loop.backward()
    grads = {n: p.grad.clone() for n, p in vae.named_parameters()}
    vae.zero_grad()
    cursor[0] = 0
    batch = beer.accumulate_elbo(vae, (X, lengths), datasize=N, nsamples=nsamp)
    assert abs(float(batch) - float(loop)) <= 1e-10 * abs(float(loop))
    batch.backward()
    for n, p in vae.named_parameters():
        assert_close(npy(p.grad), npy(grads[n]), 1e-8, 'grad ' + n)
    for p in vae.bayesian_parameters():
        assert_close(npy(batch._acc_stats[p]), npy(loop._acc_stats[p]), 1e-9, 'acc')


@pytest.mark.parametrize('T,Q,S,G', [(20001, 932, 24, 4), (8192, 600, 64, 1), (12000, 2162, 7, 16)])
def test_large_dense_products_on_the_matrix_cores(T, Q, S, G):
    '''float32 statistics-in products with a long statistics dimension (full-
    covariance latent) run on the matrix cores (dense.hip: gemm3_kernel, bf16x3);
    same results, to float32 accuracy, as the fp64 kernels.'''
    from beer_amd import _hip, kernels
    from gpu_helpers import DEV
    torch.manual_seed(4)
    K = S * G
    st = torch.randn(T, Q, dtype=torch.float64, device=DEV)
    E = torch.randn(K, Q, dtype=torch.float64, device=DEV) / Q ** .5
    w = torch.rand(T, K, dtype=torch.float64, device=DEV)
    sr = torch.rand(T, S, dtype=torch.float64, device=DEV)
    g = torch.rand(T, dtype=torch.float64, device=DEV) + .5
    llh64 = kernels.dense_llh(st, E, 30)
    llh32 = kernels.dense_llh(st.float(), E.float(), 30)
    torch.testing.assert_close(llh32.double(), llh64, rtol=0, atol=2e-6 * float(llh64.abs().max()))
    for grad in (g, None):
        b64 = kernels._llh_backward(w, grad, E)
        b32 = kernels._llh_backward(w.float(), None if grad is None else grad.float(), E.float())
        torch.testing.assert_close(b32.double(), b64, rtol=0, atol=2e-6 * float(b64.abs().max()))
    for state in (sr, None):
        a64 = kernels.dense_accumulate(st, w, state, S, G)
        a32 = kernels.dense_accumulate(st.float(), w.float(), None if state is None else state.float(),
                                       S, G)
        torch.testing.assert_close(a32, a64, rtol=0, atol=2e-6 * float(a64.abs().max()))
        # accumulation adds to what is there
        again = kernels.dense_accumulate(st.float(), w.float(),
                                         None if state is None else state.float(), S, G, acc=a32.clone())
        torch.testing.assert_close(again, 2. * a32, rtol=1e-12, atol=0)


@pytest.mark.parametrize('cov,D,ns,dtype,tol', [
    ('full', 64, 1, torch.float32, 2e-6), ('full', 40, 3, torch.float32, 2e-6),
    ('full', 8, 2, torch.float64, 1e-13), ('full', 33, 1, torch.float64, 1e-13),
    ('full', 5, 2, torch.float64, 1e-13), ('diagonal', 40, 2, torch.float64, 1e-13)])
def test_statistics_gradient_against_autograd(cov, D, ns, dtype, tol):
    '''d (mean over ns samples of phi(x)) / dx contracted with an upstream gradient:
    the kernels (wave-per-frame for full covariance with D >= 8) against torch
    autograd on the dense construction of the same statistics (fp64).'''
    from beer_amd import kernels
    from gpu_helpers import DEV
    torch.manual_seed(9)
    T = 301
    x64 = torch.randn(T * ns, D, dtype=torch.float64, device=DEV)
    ones = torch.ones(T * ns, 1, dtype=torch.float64, device=DEV)
    xr = x64.clone().requires_grad_(True)
    if cov == 'full':
        dense = torch.cat([xr, -.5 * (xr[:, :, None] * xr[:, None, :]).reshape(T * ns, D * D),
                           -.5 * ones, .5 * ones], dim=1)
    else:
        dense = torch.cat([xr, -.5 * xr ** 2, -.5 * ones, .5 * ones], dim=1)
    mean = dense.view(T, ns, -1).mean(1)
    up = torch.randn(T, mean.shape[1], dtype=torch.float64, device=DEV)
    (mean * up).sum().backward()
    mean = mean.detach()
    data = x64.to(dtype).clone().requires_grad_(True)
    stats = kernels.differentiable_stats(data, cov, ns)
    torch.testing.assert_close(stats.detach().double(), mean, rtol=0,
                               atol=tol * float(mean.abs().max()) + 1e-30)
    (stats * up.to(dtype)).sum().backward()
    torch.testing.assert_close(data.grad.double(), xr.grad, rtol=0,
                               atol=tol * float(xr.grad.abs().max()))


# ---- one sample per frame: the prior on the frame kernels (csrc/sample_grad.hip) ----------

def _no_dense_statistics(monkeypatch):
    'The [T, Q] tensor must not be formed on this route.'
    from beer_amd import kernels

    def refuse(*a, **k):
        raise AssertionError('dense [T, Q] statistics formed on the one-sample route')
    monkeypatch.setattr(kernels, 'differentiable_stats', refuse)
    monkeypatch.setattr(kernels, 'dense_llh', refuse)
    monkeypatch.setattr(kernels, 'dense_accumulate', refuse)


@pytest.mark.parametrize('name,kind', [
    ('g18_onesample_gmm_full', 'gmm'), ('g18_onesample_gmm_diagonal', 'gmm'),
    ('g18_onesample_gmm_isotropic', 'gmm'), ('g18_onesample_hmm_full', 'hmm'),
    ('g18_onesample_hmm_diagonal', 'hmm'), ('g18_onesample_hmm_isotropic', 'hmm'),
    ('g18_onesample_normal_full', 'normal')])
def test_one_sample_prior_against_reference(name, kind, monkeypatch):
    '''A prior over statistics that are phi(z_t) of differentiable samples (vae.py:63-86 with
    one sample per frame): value, gradient w.r.t. the samples and accumulated statistics of
    the reference, from the frame kernels + `beer_frames_llh_backward`.'''
    from beer_amd import kernels
    from gpu_helpers import npy, params_of, tt
    g = load_golden(name)
    prior = _prior(g, kind)
    _no_dense_statistics(monkeypatch)
    z = tt(g['z']).requires_grad_(True)
    T, ns, Dz = z.shape
    assert ns == 1
    c = tt(g['c'])
    stats = kernels.sample_stats(z.view(-1, Dz), name.rsplit('_', 1)[1])
    assert stats.source is not None and stats.detach().source is None
    exp_llh = prior.expected_log_likelihood(stats)
    assert_close(npy(exp_llh).reshape(g['exp_llh'].shape), g['exp_llh'], TOL, 'exp_llh')
    (c * exp_llh.reshape(g['exp_llh'].shape)).sum().backward()
    assert_close(npy(z.grad), g['grad_z'], TOL, 'd/dz')
    acc = prior.accumulate(stats.detach())
    for i, p in enumerate(params_of(prior)):
        assert_close(npy(acc[p]).reshape(g[f'acc.p{i}'].shape), g[f'acc.p{i}'], TOL, f'acc p{i}')


@pytest.mark.parametrize('name,kind,width', [('g18_hmm_vae_step_full', 'hmm', 16),
                                             ('g18_gmm_vae_step_full', 'gmm', 8)])
def test_vae_step_with_one_sample_against_reference(name, kind, width, monkeypatch):
    '''One ELBO + backward of a VAE with a full-covariance HMM / GMM prior and ONE sample per
    frame, the reference's weights and noise: value, gradient of every network weight,
    accumulated statistics -- without the [T, Q] statistics.'''
    import beer_amd as beer
    from beer_amd.dists import normaldiag
    from gpu_helpers import build_hmm, build_mixture, npy, params_of, tt
    g = load_golden(name)
    X = tt(g['X'])
    Dx, Dz = X.shape[1], g['noise'].shape[-1]
    enc = beer.nnet.ResidualFeedForwardNet(dim_in=Dx, nblocks=2, block_width=width)
    dec = beer.nnet.ResidualFeedForwardNet(dim_in=Dz, nblocks=2, block_width=width)
    prior = build_hmm(g) if kind == 'hmm' else build_mixture(g)
    vae = beer.VAE(prior, enc, dec, reference_broadcast=True).double().to('cuda')
    assert not vae.dense_statistics
    with torch.no_grad():
        for pname, p in vae.named_parameters():
            p.copy_(tt(g['nn.' + pname]))
    monkeypatch.setattr(normaldiag, '_randn', lambda *a, **k: tt(g['noise']))
    _no_dense_statistics(monkeypatch)
    elbo = beer.evidence_lower_bound(vae, X, nsamples=1, datasize=int(g['datasize']))
    assert abs(float(elbo) - float(g['elbo'])) <= 1e-9 * abs(float(g['elbo']))
    elbo.backward()
    for pname, p in vae.named_parameters():
        assert_close(npy(p.grad), g['nngrad.' + pname], 1e-7, 'grad ' + pname)
    for i, p in enumerate(params_of(vae)):
        assert_close(npy(elbo._acc_stats[p]).reshape(g[f'acc.p{i}'].shape), g[f'acc.p{i}'],
                     1e-9, f'acc p{i}')


@pytest.mark.parametrize('cov,T,D,K,dtype,tol', [
    ('full', 20001, 64, 120, torch.float32, 2e-6), ('full', 9000, 40, 7, torch.float32, 2e-6),
    ('full', 4096, 24, 33, torch.float32, 2e-6), ('full', 5000, 13, 4, torch.float32, 2e-6),
    ('full', 700, 64, 9, torch.float32, 2e-6), ('full', 5000, 72, 5, torch.float32, 2e-6),
    ('diagonal', 6000, 64, 120, torch.float32, 2e-6), ('isotropic', 6000, 40, 12, torch.float32, 2e-6),
    ('diagonal', 20001, 40, 33, torch.float32, 2e-6), ('diagonal', 4100, 13, 7, torch.float32, 2e-6),
    ('diagonal', 5000, 100, 64, torch.float32, 2e-6), ('isotropic', 5000, 128, 5, torch.float32, 2e-6),
    ('diagonal', 5000, 130, 5, torch.float32, 2e-6), ('diagonal', 900, 64, 12, torch.float32, 2e-6),
    ('full', 3000, 33, 6, torch.float64, 1e-12), ('diagonal', 3000, 40, 6, torch.float64, 1e-12),
    ('isotropic', 300, 5, 3, torch.float64, 1e-12)])
def test_frames_gradient_against_autograd(cov, T, D, K, dtype, tol):
    '''`beer_frames_llh_backward` (matrix cores for float32 and large T -- full covariance up to
    64 dimensions, diagonal / isotropic up to 128 --, the two dense steps over chunks or the
    generic kernel otherwise) against torch autograd in fp64 on the dense construction
    g_t sum_k w_tk phi(x_t) . E_k, with and without the per-frame factor.'''
    from beer_amd import _hip, kernels
    from gpu_helpers import DEV
    torch.manual_seed(11)
    x64 = torch.randn(T, D, dtype=torch.float64, device=DEV)
    Q = {'full': D * D + D + 2, 'diagonal': 2 * D + 2, 'isotropic': D + 3}[cov]
    E = torch.randn(K, Q, dtype=torch.float64, device=DEV) / D ** .5
    w = torch.rand(T, K, dtype=torch.float64, device=DEV)
    g = torch.rand(T, dtype=torch.float64, device=DEV) + .5
    # the workspace says which route a shape takes: none below 4096 frames (a thread per
    # output), the parameters' fragment image (25 / 7 KiB per component) on the matrix cores,
    # the same for [E1 | E2] of diagonal / isotropic Gaussians, a chunk of [frames, Q] gradients
    # for the two-step route
    nbytes = _hip.lib().beer_frames_llh_backward_workspace_bytes(
        _hip.dtype_code(dtype), _hip.COV_CODE[cov], T, D, K)
    if T < 4096:
        assert nbytes == 0
    elif dtype == torch.float32 and cov == 'full' and 8 <= D <= 64:
        assert nbytes == K * (25 if D > 32 else 7) * 1024
    elif dtype == torch.float32 and cov != 'full' and D <= 128:
        tiles = 2 if D <= 32 else 3 if D <= 48 else 4 if D <= 64 else 8
        assert nbytes == -(-K // 32) * 2 * tiles * 3 * 1024
    else:
        assert nbytes == min(T, max(1024, (256 << 20) // (Q * x64.to(dtype).element_size()))) * \
            Q * x64.to(dtype).element_size()
    for grad in (g, None):
        xr = x64.clone().requires_grad_(True)
        ones = torch.ones(T, 1, dtype=torch.float64, device=DEV)
        if cov == 'full':
            quad = -.5 * (xr[:, :, None] * xr[:, None, :]).reshape(T, D * D)
        elif cov == 'diagonal':
            quad = -.5 * xr ** 2
        else:
            quad = -.5 * (xr ** 2).sum(1, keepdim=True)
        dense = torch.cat([xr, quad, -.5 * ones, .5 * ones], dim=1)
        value = ((dense @ E.t()) * w).sum(1)
        (value if grad is None else value * grad).sum().backward()
        stats = kernels.sample_stats(x64.to(dtype), cov)
        out = kernels.frames_llh_backward(stats, w.to(dtype), None if grad is None else grad.to(dtype),
                                          E.to(dtype))
        assert out.dtype == dtype and tuple(out.shape) == (T, D)
        torch.testing.assert_close(out.double(), xr.grad, rtol=0,
                                   atol=tol * float(xr.grad.abs().max()))


def test_hmm_vae_routes_agree_on_a_large_minibatch(monkeypatch):
    '''An HMM-VAE with a full-covariance prior at config 4's latent dimension on 20 k float32
    frames in utterances: the one-sample route (frame kernels on the matrix cores,
    `sgrad_kernel`) and the dense-statistics route give the same ELBO, network gradients
    and accumulated statistics to float32 accuracy.'''
    import beer_amd as beer
    from beer_amd.dists import normaldiag
    from gpu_helpers import npy
    torch.manual_seed(3)
    Dx, Dz, S = 20, 64, 12
    lengths = [400] * 50
    T = sum(lengths)
    X = torch.randn(T, Dx, device='cuda')
    noise = torch.randn(T, 1, Dz, device='cuda')
    graph = beer.graph.Graph()
    s0, s1 = graph.add_state(), graph.add_state()
    graph.start_state, graph.end_state = s0, s1
    st = [graph.add_state(pdf_id=i) for i in range(S)]
    graph.add_arc(s0, st[0])
    for i, s in enumerate(st):
        graph.add_arc(s, s)
        graph.add_arc(s, st[(i + 1) % S])
    graph.add_arc(st[-1], s1)
    graph.normalize()
    ns = beer.NormalSet.create(torch.zeros(Dz), torch.ones(Dz), size=S, cov_type='full',
                               noise_std=.5)
    vae = beer.VAE(beer.HMM.create(graph.compile(), ns),
                   beer.nnet.ResidualFeedForwardNet(Dx, 1, 32),
                   beer.nnet.ResidualFeedForwardNet(Dz, 1, 32)).to('cuda')
    monkeypatch.setattr(normaldiag, '_randn', lambda *a, **k: noise)
    results = []
    for dense in (True, False):
        vae.dense_statistics = dense
        vae.zero_grad()
        elbo = beer.accumulate_elbo(vae, (X, lengths), datasize=10 * T)
        elbo.backward()
        results.append((float(elbo), {n: p.grad.clone() for n, p in vae.named_parameters()},
                        {p: v.clone() for p, v in elbo._acc_stats.items()}))
    (e_d, g_d, a_d), (e_s, g_s, a_s) = results
    assert abs(e_s - e_d) <= 2e-6 * abs(e_d)
    for n in g_d:
        torch.testing.assert_close(g_s[n], g_d[n], rtol=0,
                                   atol=2e-5 * float(g_d[n].abs().max()) + 1e-12)
    for p in a_d:
        assert_close(npy(a_s[p]), npy(a_d[p]), 2e-5, 'acc')


@pytest.mark.parametrize('cov,D,K', [('full', 64, 120), ('full', 40, 48), ('diagonal', 64, 120),
                                     ('full', 24, 128), ('isotropic', 40, 3), ('full', 64, 130)])
def test_per_state_log_likelihoods_on_the_matrix_cores(cov, D, K):
    '''`NormalSet.expected_log_likelihood` of float32 frames -- the per-state log-likelihoods of
    an HMM with one Gaussian per state, K mixtures of ONE component to the E-step kernel (a
    chunk of 4 / 8 component tiles up to 128 Gaussians, of 16 beyond) -- against the float64
    kernels on the same frames, to float32 accuracy.'''
    import beer_amd as beer
    from gpu_helpers import DEV
    torch.manual_seed(21)
    T = 20000
    X = torch.randn(T, D, dtype=torch.float64, device=DEV) * 1.5 + .3
    ns = beer.NormalSet.create(torch.zeros(D, dtype=torch.float64), torch.ones(D, dtype=torch.float64),
                               size=K, cov_type=cov, noise_std=1.).to(DEV)
    l64 = ns.expected_log_likelihood(ns.sufficient_statistics(X))
    ns32 = ns.float()
    l32 = ns32.expected_log_likelihood(ns32.sufficient_statistics(X.float()))
    assert l32.dtype == torch.float32 and tuple(l32.shape) == (T, K)
    # (the float32 model's parameters are the float64 ones rounded: the band of that
    # rounding, a few 1e-7 of the largest term of a logit, plus the arithmetic's)
    scale = float((X.abs().max() ** 2) * D)
    torch.testing.assert_close(l32.double(), l64, rtol=2e-6, atol=2e-6 * scale)


def _small_phone_loop(n_phones, dim, cov, dtype):
    'beer hmm mkphones / mkphoneloopgraph / mkphoneloop in memory: 3 emitting states per phone.'
    import beer_amd as beer
    units, pdf = {}, 0
    for p in range(n_phones):
        g = beer.graph.Graph()
        for sid in range(5):
            g.add_state(pdf_id=None if sid in (0, 4) else pdf + sid - 1)
        g.start_state, g.end_state = 0, 4
        for arc in ((0, 1), (1, 1), (1, 2), (2, 2), (2, 3), (3, 3), (3, 4)):
            g.add_arc(*arc)
        g.normalize()
        units[p] = g
        pdf += 3
    graph = beer.graph.Graph()
    graph.start_state, graph.end_state = graph.add_state(), graph.add_state()
    pivot = graph.add_state()
    u2s = {p: graph.add_state() for p in units}
    graph.add_arc(graph.start_state, pivot)
    graph.add_arc(pivot, graph.end_state)
    for p in units:
        graph.add_arc(pivot, u2s[p])
        graph.add_arc(u2s[p], pivot)
    graph.normalize()
    for p, hmm in units.items():
        graph.replace_state(u2s[p], hmm)
    graph.normalize()
    ns = beer.NormalSet.create(torch.zeros(dim, dtype=dtype), torch.ones(dim, dtype=dtype),
                               size=3 * n_phones, cov_type=cov, noise_std=1.)
    return beer.PhoneLoop.create(graph.compile(), {p: 3 * p for p in units},
                                 {p: 3 * p + 2 for p in units}, ns)


@pytest.mark.parametrize('nsamp', [1, 2])
def test_phone_loop_vae_batch_equals_per_utterance_loop(nsamp, monkeypatch):
    '''A VAE whose prior is a phone loop (config 4's model): accumulate_elbo on a minibatch of
    utterances == the reference's loop of per-utterance calls -- value, network gradients, the
    Gaussians' statistics AND the phone counts, which take the first frame of EVERY utterance
    (phoneloop.py:88-95).  One sample per frame (frame kernels) and two (dense statistics).'''
    import beer_amd as beer
    from beer_amd.dists import normaldiag
    from gpu_helpers import npy
    torch.manual_seed(6)
    Dx, Dz = 5, 4
    lengths = [30, 17, 44, 25]
    X = torch.randn(sum(lengths), Dx, dtype=torch.float64, device='cuda')
    master = torch.randn(sum(lengths), nsamp, Dz, dtype=torch.float64, device='cuda')
    prior = _small_phone_loop(3, Dz, 'full', torch.float64)
    vae = beer.VAE(prior, beer.nnet.ResidualFeedForwardNet(Dx, 1, 8),
                   beer.nnet.ResidualFeedForwardNet(Dz, 1, 8)).double().to('cuda')
    cursor = [0]

    def fake_randn(n, *rest, **conf):
        out = master[cursor[0]:cursor[0] + n]
        cursor[0] += n
        return out
    monkeypatch.setattr(normaldiag, '_randn', fake_randn)
    N = 1000
    loop = beer.evidence_lower_bound(datasize=N)
    for u in torch.split(X, lengths):
        loop += beer.evidence_lower_bound(vae, u, datasize=N, nsamples=nsamp)
    loop.backward()
    grads = {n: p.grad.clone() for n, p in vae.named_parameters()}
    vae.zero_grad()
    cursor[0] = 0
    batch = beer.accumulate_elbo(vae, (X, lengths), datasize=N, nsamples=nsamp)
    assert abs(float(batch) - float(loop)) <= 1e-10 * abs(float(loop))
    batch.backward()
    for n, p in vae.named_parameters():
        assert_close(npy(p.grad), npy(grads[n]), 1e-8, 'grad ' + n)
    params = list(vae.bayesian_parameters())
    assert len(params) == 2                      # the Gaussians and the phone weights
    for p in params:
        assert_close(npy(batch._acc_stats[p]), npy(loop._acc_stats[p]), 1e-9, 'acc')


@pytest.mark.parametrize('cov,T,D,K,dtype,tol', [('full', 6000, 24, 12, torch.float32, 1e-5),
                                                 ('diagonal', 6000, 40, 33, torch.float32, 1e-5),
                                                 ('full', 500, 9, 5, torch.float64, 1e-9),
                                                 ('isotropic', 500, 9, 5, torch.float64, 1e-9)])
def test_one_sample_gmm_prior_against_the_oracle(cov, T, D, K, dtype, tol):
    '''A GMM prior over phi(z_t) of seeded samples: value, gradient w.r.t. the samples and
    accumulated statistics of the HIP path against the CPU oracle (oracle/beer_oracle.py:
    vae_gmm_prior, prior_gradient_wrt_samples -- pinned on the reference's G18 goldens).'''
    import beer_amd as beer
    from beer_amd import kernels
    from gpu_helpers import DEV, npy, params_of
    from helpers import orc
    torch.manual_seed(31)
    Z = (torch.randn(T, D, dtype=torch.float64) * 1.3 + .2).to(dtype)
    c = torch.rand(T, dtype=torch.float64).to(dtype) + .5
    ns = beer.NormalSet.create(torch.zeros(D, dtype=dtype), torch.ones(D, dtype=dtype), size=K,
                               cov_type=cov, noise_std=1.)
    prior = beer.Mixture.create(ns).to(DEV)
    p0, p1 = params_of(prior)
    as64 = lambda d: [npy(getattr(d.params, n)).astype(np.float64) for n in d._std_params_def]
    Zn = Z.numpy().astype(np.float64)
    value, resps, exp_T = orc.vae_gmm_prior(cov, Zn, as64(p0.posterior), as64(p1.posterior)[0])
    grad = orc.prior_gradient_wrt_samples(cov, Zn, resps, exp_T, c.numpy().astype(np.float64))
    z = Z.to(DEV).requires_grad_(True)
    stats = kernels.sample_stats(z, cov)
    got = prior.expected_log_likelihood(stats)
    assert_close(npy(got), value, tol, 'value')
    (c.to(DEV) * got).sum().backward()
    assert_close(npy(z.grad), grad, 10 * tol, 'd/dz')
    acc = prior.accumulate(stats.detach())[p0]
    assert_close(npy(acc), resps.T @ orc.SUFFSTATS[cov](Zn), 10 * tol, 'acc')


def test_exact_mode_keeps_the_sample_gradient_off_the_bf16_kernels():
    '''`set_f32_mode('exact')`: the gradient w.r.t. the samples comes from the float64-accumulating
    kernel (no workspace handed over) -- equal to the fp64 result to float32 rounding of the
    output, and different in the last bits from the bf16x3 kernel's.'''
    from beer_amd import _hip, kernels
    from gpu_helpers import DEV
    torch.manual_seed(12)
    T, D, K = 5000, 16, 6
    X = torch.randn(T, D, device=DEV)
    E = torch.randn(K, D * D + D + 2, device=DEV) / D ** .5
    w = torch.rand(T, K, device=DEV)
    st = kernels.sample_stats(X, 'full')
    fast = kernels.frames_llh_backward(st, w, None, E)
    with _hip.exact_f32():
        exact = kernels.frames_llh_backward(st, w, None, E)
    ref = kernels.frames_llh_backward(kernels.sample_stats(X.double(), 'full'), w.double(), None,
                                      E.double())
    scale = float(ref.abs().max())
    assert float((exact.double() - ref).abs().max()) <= 1e-7 * scale
    assert float((fast.double() - ref).abs().max()) <= 2e-6 * scale
    assert not torch.equal(fast, exact)


@pytest.mark.parametrize('cov', ['full', 'diagonal'])
def test_sample_gradient_at_config4_size(cov):
    '''`beer_frames_llh_backward` at BASELINE config 4's own size (1 M samples of a 64-dimensional
    latent variable, 120 states) through size-independent properties: it is linear in the
    posteriors and in the per-frame factor, rows depend on their own frame only (any slice
    of the batch reproduces its rows bit for bit on the same route), and sampled rows agree
    with the float64 kernel.'''
    from beer_amd import kernels
    from gpu_helpers import DEV
    torch.manual_seed(44)
    T, D, K = 1_000_000, 64, 120
    Q = D * D + D + 2 if cov == 'full' else 2 * D + 2
    X = torch.randn(T, D, device=DEV)
    E = torch.randn(K, Q, device=DEV) / D ** .5
    w1 = torch.rand(T, K, device=DEV)
    w2 = torch.rand(T, K, device=DEV)
    g = torch.rand(T, device=DEV) + .5
    st = kernels.sample_stats(X, cov)
    a = kernels.frames_llh_backward(st, w1, None, E)
    b = kernels.frames_llh_backward(st, w2, None, E)
    ab = kernels.frames_llh_backward(st, w1 + w2, None, E)
    scale = float(ab.abs().max())
    assert float((ab - (a + b)).abs().max()) <= 4e-6 * scale           # linear in the posteriors
    ag = kernels.frames_llh_backward(st, w1, g, E)
    assert float((ag - g[:, None] * a).abs().max()) <= 1e-6 * scale    # ... and in the factor
    lo, n = 123_456, 65_536 + 17
    part = kernels.frames_llh_backward(kernels.sample_stats(X[lo:lo + n], cov), w1[lo:lo + n],
                                       None, E)
    assert torch.equal(part, a[lo:lo + n])                             # rows are independent
    idx = torch.randint(0, T, (3000,), device=DEV)
    ref = kernels.frames_llh_backward(kernels.sample_stats(X[idx].double(), cov), w1[idx].double(),
                                      None, E.double())
    assert float((a[idx].double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize('cov', ['full', 'diagonal'])
def test_vae_with_mixture_states_routes_agree(cov, monkeypatch):
    '''A VAE whose prior is an HMM with a MIXTURE per state (mixtureset.py:85-98: the
    log-normalisers are detached, so the prior sends no gradient to the samples): one sample per
    frame on the frame kernels against the dense-statistics route, fp64 -- value, network
    gradients, statistics of the Gaussians and of the mixture weights.'''
    import beer_amd as beer
    from beer_amd.dists import normaldiag
    from gpu_helpers import npy
    torch.manual_seed(8)
    Dx, Dz, S, G = 5, 4, 3, 4
    lengths = [40, 33, 57]
    T = sum(lengths)
    X = torch.randn(T, Dx, dtype=torch.float64, device='cuda')
    noise = torch.randn(T, 1, Dz, dtype=torch.float64, device='cuda')
    graph = beer.graph.Graph()
    s0, s1 = graph.add_state(), graph.add_state()
    graph.start_state, graph.end_state = s0, s1
    st = [graph.add_state(pdf_id=i) for i in range(S)]
    graph.add_arc(s0, st[0])
    for i, s in enumerate(st):
        graph.add_arc(s, s)
        graph.add_arc(s, st[(i + 1) % S])
    graph.add_arc(st[-1], s1)
    graph.normalize()
    ns = beer.NormalSet.create(torch.zeros(Dz, dtype=torch.float64),
                               torch.ones(Dz, dtype=torch.float64), size=S * G, cov_type=cov,
                               noise_std=1.)
    prior = beer.HMM.create(graph.compile(), beer.MixtureSet.create(S, ns))
    vae = beer.VAE(prior, beer.nnet.ResidualFeedForwardNet(Dx, 1, 8),
                   beer.nnet.ResidualFeedForwardNet(Dz, 1, 8)).double().to('cuda')
    monkeypatch.setattr(normaldiag, '_randn', lambda *a, **k: noise)
    results = []
    for dense in (True, False):
        vae.dense_statistics = dense
        vae.zero_grad()
        elbo = beer.accumulate_elbo(vae, (X, lengths), datasize=10 * T)
        elbo.backward()
        results.append((float(elbo), {n: p.grad.clone() for n, p in vae.named_parameters()},
                        {p: v.clone() for p, v in elbo._acc_stats.items()}))
    (e_d, g_d, a_d), (e_s, g_s, a_s) = results
    assert abs(e_s - e_d) <= 1e-10 * abs(e_d)
    for n in g_d:
        assert_close(npy(g_s[n]), npy(g_d[n]), 1e-9, 'grad ' + n)
    assert len(a_d) == 2
    for p in a_d:
        assert_close(npy(a_s[p]), npy(a_d[p]), 1e-9, 'acc')


@pytest.mark.parametrize('cov,S,D,nutt', [('full', 120, 64, 56), ('diagonal', 120, 64, 56)])
def test_one_sample_hmm_prior_at_config4_shape_against_the_oracle(cov, S, D, nutt):
    '''the float32 one-sample route of an HMM prior at BASELINE
    config 4's shape -- a 40-phone x 3-state loop (S = 120 single Gaussians), 64-d latent,
    > 16 384 float32 samples in ragged utterances, so `llhx_kernel`, the fused
    forward-backward launch and `sgrad_kernel` / `sgrad_diag_kernel` (bf16x3 on the matrix
    cores) are what runs -- against the CPU oracle's fp64 restatement of the reference's
    chain (oracle/beer_oracle.py: `vae_hmm_prior`, `prior_gradient_wrt_samples`, pinned on
    the reference's G18 goldens; vae.py:63-86, hmm.py:73-92, normalwishart.py:88-92): per-frame
    value, state posteriors at the pdf ids, gradient w.r.t. the samples and accumulated
    statistics, float32 at 1e-5 (posteriors: inside the band of the oracle's own float32 run).'''
    import beer_amd as beer
    from beer_amd import _hip, kernels
    from gpu_helpers import DEV, npy
    from helpers import assert_stats_close, orc, rel_err
    torch.manual_seed(5)
    rng = np.random.RandomState(5)
    P = S // 3
    graph = beer.graph.Graph()
    s0, s1 = graph.add_state(), graph.add_state()
    graph.start_state, graph.end_state = s0, s1
    pivot = graph.add_state()
    graph.add_arc(s0, pivot)
    graph.add_arc(pivot, s1)
    for p in range(P):
        st = [graph.add_state(pdf_id=3 * p + k) for k in range(3)]
        graph.add_arc(pivot, st[0])
        for k in range(3):
            graph.add_arc(st[k], st[k], .75)
            graph.add_arc(st[k], st[k + 1] if k < 2 else pivot, .25)
    graph.normalize()
    cg = graph.compile()
    ns = beer.NormalSet.create(torch.zeros(D), torch.ones(D), size=S, cov_type=cov, noise_std=1.)
    prior = beer.HMM.create(cg, ns).to(DEV)
    p0 = list(prior.bayesian_parameters())[0]
    mu = npy(p0.posterior.params.mean).astype(np.float64)
    lengths = [int(n) for n in rng.randint(250, 350, nutt)]
    T = sum(lengths)
    assert T >= _hip.FAST_MIN_FRAMES
    parts = []
    for n in lengths:
        seq = np.repeat(rng.randint(0, P, n // 25 + 1), 25)[:n]
        parts.append(mu[3 * seq + rng.randint(0, 3, n)] + rng.randn(n, D) * 1.2)
    Z32 = np.concatenate(parts).astype(np.float32)
    Zn = Z32.astype(np.float64)
    c_up = (rng.rand(T) + .5).astype(np.float32)
    post = [npy(getattr(p0.posterior.params, n)).astype(np.float64) for n in p0.posterior._std_params_def]
    og = dict(init=npy(cg.init_log_probs).astype(np.float64), final=npy(cg.final_log_probs).astype(np.float64),
              trans=npy(cg.trans_log_probs).astype(np.float64), order=np.asarray(cg.pdf_id_mapping))

    def oracle(dtype):
        vals, resps, off = [], [], 0
        og_t = {k: (v.astype(dtype) if k != 'order' else v) for k, v in og.items()}
        with np.errstate(invalid='ignore', divide='ignore'):
            for n in lengths:
                v, r, e = orc.vae_hmm_prior(cov, Zn[off:off + n].astype(dtype),
                                            [a.astype(dtype) for a in post], og_t)
                vals.append(v)
                resps.append(r)
                off += n
        return np.concatenate(vals), np.concatenate(resps), e
    value, resps, exp_T = oracle(np.float64)
    value32, resps32, _ = oracle(np.float32)
    grad = orc.prior_gradient_wrt_samples(cov, Zn, resps, exp_T, c_up.astype(np.float64))
    z = torch.from_numpy(Z32).to(DEV).requires_grad_(True)
    stats = kernels.sample_stats(z, cov)
    got = prior.expected_log_likelihood(stats, utt_lengths=lengths)
    band = lambda ref32, truth: max(1e-5, 1.5 * rel_err(ref32.astype(np.float64), truth))   # noqa: E731
    assert rel_err(npy(got).astype(np.float64), value) <= band(value32, value), 'per-frame value'
    sr = npy(prior.cache['scaled_pdf_resps']).astype(np.float64)
    assert rel_err(sr, resps) <= band(resps32, resps), 'state posteriors'
    assert abs(float(got.detach().double().sum()) - value.sum()) <= 1e-5 * abs(value.sum())
    (torch.from_numpy(c_up).to(DEV) * got).sum().backward()
    # the gradient of the HIP path multiplies ITS posteriors: hold it against the oracle's
    # gradient at 1e-5 of the largest entry plus what the posteriors' own float32 band moves
    g_band = max(1e-5, 1.5 * rel_err(orc.prior_gradient_wrt_samples(
        cov, Zn, resps32.astype(np.float64), exp_T, c_up.astype(np.float64)), grad))
    assert rel_err(npy(z.grad).astype(np.float64), grad) <= g_band, 'd/dz'
    acc = prior.accumulate(stats.detach())[p0]
    acc_band = max(1e-5, 1.5 * rel_err(resps32.astype(np.float64).T @ orc.SUFFSTATS[cov](Zn),
                                       resps.T @ orc.SUFFSTATS[cov](Zn)))
    assert_stats_close(npy(acc), resps.T @ orc.SUFFSTATS[cov](Zn), D, acc_band, 'acc')


@pytest.mark.gpu
def test_row_split_layers_on_the_gpu_match_nn_linear():
    '''`beer_amd.nnet.Linear` at a minibatch size where the split is taken (>= 65536 rows), on the
    device: the same output and input gradient as nn.Linear, the weight gradient within float32
    summation noise of the float64 result (and no farther from it than torch's own long-chain
    product), through a residual block and the VAE's heads as the config-4 step uses them.'''
    from beer_amd.nnet import linear
    from gpu_helpers import DEV
    import beer_amd as beer
    torch.manual_seed(5)
    T, fin, fout = 70_003, 40, 128
    assert T >= linear.ROW_SPLIT_MIN
    x = torch.randn(T, fin, device=DEV)
    up = torch.randn(T, fout, device=DEV)
    lin = beer.nnet.Linear(fin, fout).to(DEV)
    ref = torch.nn.Linear(fin, fout).to(DEV)
    ref.load_state_dict(lin.state_dict())
    ref64 = torch.nn.Linear(fin, fout).to(DEV).double()
    ref64.load_state_dict(lin.state_dict())
    xs = [x.clone().requires_grad_(True), x.clone().requires_grad_(True), x.double().requires_grad_(True)]
    ys = [lin(xs[0]), ref(xs[1]), ref64(xs[2])]
    assert ys[0].grad_fn.name().startswith('_RowSplitLinear') and torch.equal(ys[0], ys[1])
    for y, u in zip(ys, (up, up, up.double())):
        (y.tanh() * u).sum().backward()
    assert torch.equal(xs[0].grad, xs[1].grad)
    truth = ref64.weight.grad
    err_split = float((lin.weight.grad.double() - truth).abs().max() / truth.abs().max())
    err_torch = float((ref.weight.grad.double() - truth).abs().max() / truth.abs().max())
    assert err_split <= 2e-6 and err_split <= 2 * err_torch + 1e-7, (err_split, err_torch)
    assert float((lin.bias.grad.double() - ref64.bias.grad).abs().max() / ref64.bias.grad.abs().max()) <= 2e-6
    net = beer.nnet.ResidualFeedForwardNet(fin, 2, 64).to(DEV)
    assert all(isinstance(b.layer1, linear.Linear) and isinstance(b.layer2, linear.Linear) for b in net.blocks)
    out = net(x.requires_grad_(True))
    out.sum().backward()
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in net.parameters())
