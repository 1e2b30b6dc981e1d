"""The call sequences of the reference's example notebooks, written against a
module handle so that ONE text runs on the reference (`make_golden.py`, in the
build container: it writes g16_notebooks.npz) and on beer_amd (`tests/
test_notebooks.py`, on the GPU box: `import beer_amd as beer`).

Which calls, in which order, with which arguments: examples/Mixture Model.ipynb
(cells 5, 9, 12, 14, 16, 18, 20), examples/HMM.ipynb (cells 5-9, 13) and
examples/HMM_align.ipynb (cells 4-10, 15) of the reference.  The plotting cells
only read `model.categorical.mean`, `normal.mean`, `normal.cov` of every
component -- those reads are part of the sequences.  Data are synthetic and
seeded here (the notebooks draw them unseeded); epochs are cut from 100 to
`epochs` to keep the fixture small.
"""

import numpy as np
import torch


def _np(t):
    return t.detach().cpu().numpy().copy() if isinstance(t, torch.Tensor) else np.asarray(t)


def mixture_data(seed=0, n=200):
    'Two clusters as in the notebook (cell 3).'
    rng = np.random.RandomState(seed)
    a = rng.multivariate_normal([-5, 5], .5 * np.array([[.75, 0.], [0, 5.]]), size=n)
    b = rng.multivariate_normal([5, 5], 2 * np.array([[2, -.5], [-.5, .75]]), size=n)
    data = np.vstack([a, b])
    rng.shuffle(data)
    return data


def mixture_model(beer, data, variant, epochs, seed=1):
    '''variant 'dirichlet' (cells 5-10), 'sb' (12-16), 'sb_hyper' (18-21).'''
    X = torch.from_numpy(data).double()
    mean = torch.from_numpy(data.mean(axis=0)).double()
    var = torch.from_numpy(np.var(data, axis=0)).double()
    torch.manual_seed(seed)
    if variant == 'dirichlet':
        modelset = beer.NormalSet.create(mean, var, size=10, prior_strength=1., noise_std=1.,
                                         cov_type='full')
        model = beer.Mixture.create(modelset, prior_strength=100.)
    else:
        modelset = beer.NormalSet.create(mean, var, size=20, prior_strength=1., noise_std=1,
                                         cov_type='full')
        if variant == 'sb':
            cat = beer.SBCategorical.create(truncation=len(modelset), prior_strength=10)
        else:
            cat = beer.SBCategoricalHyperPrior.create(truncation=len(modelset),
                                                      prior_strength=10., hyper_prior_strength=1)
        model = beer.Mixture.create(modelset, categorical=cat, prior_strength=1.)
    model = model.double()
    out = {'repr_nonempty': np.array(len(repr(model)) > 0)}
    out['weights0'] = _np(model.categorical.mean)
    optim = beer.VBConjugateOptimizer(model.mean_field_factorization(), 1.)
    elbos = []
    for _ in range(epochs):
        optim.init_step()
        elbo = beer.evidence_lower_bound(model, X)
        elbo.backward()
        optim.step()
        elbos.append(float(elbo) / len(X))
    out['elbos'] = np.asarray(elbos)
    out['weights'] = _np(model.categorical.mean)
    out['means'] = np.stack([_np(normal.mean) for normal in model.modelset])
    out['covs'] = np.stack([_np(normal.cov) for normal in model.modelset])
    if variant != 'dirichlet':
        out['ordering'] = _np(model.categorical.ordering)
    if variant == 'sb_hyper':
        post = model.categorical.concentration.posterior
        out['conc_shape'], out['conc_rate'] = _np(post.params.shape), _np(post.params.rate)
    return out


def hmm_data(seed=3, nsamples=300):
    'Three-state chain with Gaussian emissions (HMM.ipynb cell 3).'
    rng = np.random.RandomState(seed)
    trans = np.array([[.5, .5, 0], [0, .5, .5], [.5, 0, .5]])
    means = [np.array([-3., 8.]), np.array([10., 10.]), np.array([1., -2.])]
    covs = [np.array([[.75, -.5], [-.5, 2.]]), np.array([[2., 1.], [1., .75]]), np.eye(2)]
    states = np.zeros(nsamples, dtype=int)
    data = np.zeros((nsamples, 2))
    data[0] = rng.multivariate_normal(means[0], covs[0])
    for n in range(1, nsamples):
        states[n] = rng.choice(3, p=trans[states[n - 1]])
        data[n] = rng.multivariate_normal(means[states[n]], covs[states[n]])
    return data, states


def _loop_graph(beer):
    'HMM.ipynb cell 5: three emitting states in a loop, every state may end.'
    graph = beer.graph.Graph()
    s0, s4 = graph.add_state(), graph.add_state()
    graph.start_state, graph.end_state = s0, s4
    s1, s2, s3 = (graph.add_state(pdf_id=i) for i in range(3))
    for a, b in ((s0, s1), (s1, s1), (s1, s2), (s2, s2), (s2, s3), (s3, s3), (s3, s1),
                 (s1, s4), (s2, s4), (s3, s4)):
        graph.add_arc(a, b)
    graph.normalize()
    return graph


def hmm(beer, data, epochs, seed=2):
    'HMM.ipynb cells 5-9, 13: three covariance types trained side by side, then decode.'
    cgraph = _loop_graph(beer).compile()
    mean = torch.from_numpy(data.mean(axis=0)).float()
    cov = torch.from_numpy(np.cov(data.T)).float()
    torch.manual_seed(seed)
    models = {}
    for name, cov_type in (('hmm_iso', 'isotropic'), ('hmm_diag', 'diagonal'), ('hmm_full', 'full')):
        modelset = beer.NormalSet.create(mean, cov, size=3, prior_strength=1., noise_std=0,
                                         cov_type=cov_type)
        models[name] = beer.HMM.create(cgraph, modelset).double()
    X = torch.from_numpy(data).double()
    optims = {name: beer.VBConjugateOptimizer(model.mean_field_factorization(), 1.)
              for name, model in models.items()}
    elbos = {name: [] for name in models}
    for _ in range(epochs):
        for name, model in models.items():
            optims[name].init_step()
            elbo = beer.evidence_lower_bound(model, X, datasize=len(X), viterbi=False)
            elbo.backward()
            elbos[name].append(float(elbo) / len(X))
            optims[name].step()
    out = {'final_log_probs': _np(cgraph.final_log_probs)}
    for name in models:
        out[f'{name}.elbos'] = np.asarray(elbos[name])
    out['best_path'] = _np(models['hmm_full'].decode(X))
    return out


def align_data(seed=4, nsamples=30):
    'A B A sequence of three-state units (HMM_align.ipynb cell 2), about 3 frames a state.'
    rng = np.random.RandomState(seed)
    means = [np.array([-1.5, 3.]), np.array([-1.5, 4.]), np.array([-1.5, 5.]),
             np.array([1., -3.]), np.array([1., -2.]), np.array([1., -1.])]
    cov_a, cov_b = np.array([[.75, -.5], [-.5, 2.]]), np.array([[2., 1.], [1., .75]])
    seq = [0, 1, 2, 3, 4, 5, 0, 1, 2]
    per = nsamples // len(seq)
    states = np.repeat(seq, per)
    states = np.concatenate([states, np.full(nsamples - len(states), seq[-1])])
    data = np.stack([rng.multivariate_normal(means[s], cov_a if s < 3 else cov_b) for s in states])
    return data, states


def hmm_align(beer, data, epochs, seed=5):
    '''HMM_align.ipynb cells 4-10, 15: a 3-state loop model and a 6-pdf model trained
    through an alignment graph that repeats pdf ids (A B A), hard (viterbi=True)
    assignments, then the state posteriors of both.'''
    graph = beer.graph.Graph()
    s0 = graph.add_state()
    s1, s2, s3 = (graph.add_state(pdf_id=i) for i in range(3))
    s4 = graph.add_state()
    graph.start_state, graph.end_state = s0, s4
    for a, b in ((s0, s1), (s1, s1), (s1, s2), (s2, s2), (s2, s3), (s3, s3), (s3, s1), (s3, s4)):
        graph.add_arc(a, b)
    graph.normalize()
    loop_graph = graph.compile()

    graph = beer.graph.Graph()
    first = graph.add_state()
    chain = [graph.add_state(pdf_id=p) for p in (0, 1, 2, 3, 4, 5, 0, 1, 2)]
    last = graph.add_state()
    graph.start_state, graph.end_state = first, last
    graph.add_arc(first, chain[0])
    for a, b in zip(chain, chain[1:] + [last]):
        graph.add_arc(a, a)
        graph.add_arc(a, b)
    graph.normalize()
    ali_graph = graph.compile().double()

    mean = torch.from_numpy(data.mean(axis=0)).float()
    cov = torch.from_numpy(np.cov(data.T)).float()
    torch.manual_seed(seed)
    modelset = beer.NormalSet.create(mean, cov, size=loop_graph.n_states, prior_strength=1.,
                                     noise_std=1., cov_type='full')
    loop = beer.HMM.create(loop_graph, modelset)
    modelset = beer.NormalSet.create(mean, cov, size=ali_graph.n_states, prior_strength=1.,
                                     noise_std=1., cov_type='full')
    align = beer.HMM.create(ali_graph, modelset)
    models = {'loop': loop.double(), 'align': align.double()}
    inf_graphs = {'loop': None, 'align': ali_graph}
    X = torch.from_numpy(data).double()
    optims = {name: beer.VBConjugateOptimizer(model.mean_field_factorization(), 1.)
              for name, model in models.items()}
    elbos = {name: [] for name in models}
    for _ in range(epochs):
        for name, model in models.items():
            optims[name].init_step()
            elbo = beer.evidence_lower_bound(model, X, datasize=len(X),
                                             inference_graph=inf_graphs[name], viterbi=True)
            elbo.backward()
            elbos[name].append(float(elbo) / len(X))
            optims[name].step()
    return {'loop.elbos': np.asarray(elbos['loop']), 'align.elbos': np.asarray(elbos['align']),
            'loop.posts': _np(models['loop'].posteriors(X)),
            'align.posts': _np(models['align'].posteriors(X, ali_graph))}


def hmm_vae_data(seed=7, nsamples=120):
    'HMM-VAE.ipynb cell 1: the three-state chain of HMM.ipynb with its own means.'
    rng = np.random.RandomState(seed)
    trans = np.array([[.5, .5, 0], [0, .5, .5], [.5, 0, .5]])
    means = [np.array([-1.5, 4.]) * 2, np.array([5., 5.]) * 2, np.array([1., -2.]) * 2]
    covs = [np.array([[.75, -.5], [-.5, 2.]]), np.array([[2., 1.], [1., .75]]), np.eye(2)]
    states = np.zeros(nsamples, dtype=int)
    data = np.zeros((nsamples, 2))
    data[0] = rng.multivariate_normal(means[0], covs[0])
    for n in range(1, nsamples):
        states[n] = rng.choice(3, p=trans[states[n - 1]])
        data[n] = rng.multivariate_normal(means[states[n]], covs[states[n]])
    return data, states


def hmm_vae(beer, data, hmm_epochs, epochs, update_prior_after_epoch, randomness, nn_init=None,
            seed=8):
    '''HMM-VAE.ipynb cells 2-5 (the full-covariance HMM, trained alone first) and 7-9 (a VAE
    around it: residual encoder / decoder of width 2, `VBOptimizer(VBConjugateOptimizer(lrate=0),
    Adam)`, `evidence_lower_bound(vae, X, nsamples=5)`, the prior's learning rate switched on
    after `update_prior_after_epoch` epochs).  `randomness` is the caller's context manager
    around the epochs: the reference side records the noise `posts.sample` draws, the
    replay feeds it back.  `nn_init` (name -> array): initial network weights to load (the
    replay takes the reference's; None: keep the seeded initialisation and report it).'''
    cgraph = _loop_graph(beer).compile()
    data_mean = torch.from_numpy(data.mean(axis=0)).float()
    data_var = torch.from_numpy(np.cov(data.T)).float()
    torch.manual_seed(seed)
    modelset = beer.NormalSet.create(data_mean, data_var, size=3, prior_strength=1., noise_std=0,
                                     cov_type='full')
    hmm_full = beer.HMM.create(cgraph, modelset).double()
    X = torch.from_numpy(data)
    optim = beer.VBConjugateOptimizer(hmm_full.mean_field_factorization(), 1.)
    hmm_elbos = []
    for _ in range(hmm_epochs):
        optim.init_step()
        elbo = beer.evidence_lower_bound(hmm_full, X, datasize=len(X), viterbi=False)
        elbo.backward()
        hmm_elbos.append(float(elbo) / len(X))
        optim.step()
    # cell 7
    encoder = beer.nnet.ResidualFeedForwardNet(dim_in=2, nblocks=2, block_width=2).double()
    decoder = beer.nnet.ResidualFeedForwardNet(dim_in=2, nblocks=2, block_width=2).double()
    vae = beer.VAE(hmm_full, encoder, decoder).double()
    out = {'hmm.elbos': np.asarray(hmm_elbos)}
    if nn_init is not None:
        with torch.no_grad():
            for name, p in vae.named_parameters():
                p.copy_(torch.from_numpy(np.asarray(nn_init[name])).to(p.device))
    else:
        for name, p in vae.named_parameters():
            out['nn_init.' + name] = _np(p)
    # cell 8
    prior_lrate = 1.
    cjg_optim = beer.VBConjugateOptimizer(vae.mean_field_factorization(), lrate=0)
    std_optim = torch.optim.Adam(vae.parameters(), lr=1e-3)
    optim = beer.VBOptimizer(cjg_optim, std_optim)
    # cell 9
    elbos = []
    with randomness:
        for e in range(epochs):
            optim.init_step()
            elbo = beer.evidence_lower_bound(vae, X, nsamples=5)
            elbo.backward()
            optim.step()
            if e >= update_prior_after_epoch:
                cjg_optim.lrate = prior_lrate
            elbos.append(float(elbo) / len(X))
    out['elbos'] = np.asarray(elbos)
    for name, p in vae.named_parameters():
        out['nn_final.' + name] = _np(p)
    post = hmm_full.modelset.original_modelset.means_precisions.posterior
    out['prior.mean'], out['prior.scale_matrix'] = _np(post.params.mean), _np(post.params.scale_matrix)
    return out
