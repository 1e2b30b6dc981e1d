"""Build beer_amd models from golden arrays (GPU tests)."""

import numpy as np
import torch

import beer_amd as beer
from helpers import dist_cls, std_params

DEV = 'cuda'


def tt(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


def npy(t):
    return t.detach().cpu().numpy()


def build_dist(g, prefix, dtype=None):
    cls = getattr(beer.dists, dist_cls(g, prefix))
    return cls.from_std_parameters(*[tt(a, dtype) for a in std_params(g, prefix)])


def build_param(g, prefix, prior_prefix=None, dtype=None):
    prior_prefix = prior_prefix or prefix
    return beer.ConjugateBayesianParameter(build_dist(g, prior_prefix + '.prior', dtype),
                                           build_dist(g, prefix + '.posterior', dtype))


def build_mixture(g, prefix='init'):
    ns = beer.NormalSet(build_param(g, prefix + '.p0'))
    cat = beer.Categorical(build_param(g, prefix + '.p1'))
    return beer.Mixture(cat, ns)


def build_graph(g, prefix):
    return beer.graph.CompiledGraph(tt(g[prefix + '.init']), tt(g[prefix + '.final']),
                                    tt(g[prefix + '.trans']),
                                    [int(i) for i in g[prefix + '.pdf_id_mapping']])


def build_hmm(g, prefix='init'):
    return beer.HMM.create(build_graph(g, 'graph'), beer.NormalSet(build_param(g, prefix + '.p0')))


def build_phoneloop(g, kind, prefix='init'):
    sets, i = [], 0
    for S, G in zip(g['group_sizes'], g['group_ncomp']):
        ns = beer.NormalSet(build_param(g, f'{prefix}.p{i}'))
        cs = beer.CategoricalSet(build_param(g, f'{prefix}.p{i + 1}'))
        sets.append(beer.MixtureSet(cs, ns))
        i += 2
    emissions = beer.JointModelSet(sets)
    wparam = build_param(g, f'{prefix}.p{i}')
    if kind == 'dirichlet':
        cat = beer.Categorical(wparam)
    elif kind == 'dirichlet_process':
        cat = beer.SBCategorical(wparam)
    else:
        cat = beer.SBCategoricalHyperPrior(wparam, build_param(g, f'{prefix}.concentration'))
        # the constructor's callback recomputed prior[:, 1] = E[concentration] in
        # fp64; the reference did it in fp32 before .double(): restore its value.
        cat.stickbreaking.prior.params.concentrations.copy_(
            tt(g[f'{prefix}.p{i}.prior.concentrations']))
    start = {f'p{j}': int(v) for j, v in enumerate(g['start_idxs'])}
    end = {f'p{j}': int(v) for j, v in enumerate(g['end_idxs'])}
    ploop = beer.PhoneLoop(build_graph(g, 'graph'), emissions, start, end, cat)
    # The reference built its model in fp32 and then called .double(): the
    # loop-back transitions written by the constructor's callback carry fp32
    # rounding.  Start from exactly the reference's state.
    ploop.graph.trans_log_probs.copy_(tt(g['graph.trans']))
    return ploop, i


def params_of(model):
    return list(model.bayesian_parameters())


def check_posterior(param, g, prefix, tol, assert_close):
    for name, ref in zip(param.posterior._std_params_def, std_params(g, prefix)):
        got = npy(getattr(param.posterior.params, name)).reshape(ref.shape)
        assert_close(got, ref, tol, f'{prefix}.{name}')


def oracle_mixtureset(X, cov, ns, weights_model, S, G, state_resps=None, dtype=np.float64):
    """The numpy oracle's truth for what the mixture(-set) kernels compute from frames `X`, a
    NormalSet `ns` of S * G components and the Dirichlet posterior of `weights_model`
    (a Mixture: S = 1, or a MixtureSet): per-state log-normalisers [T, S], responsibilities
    within each state's mixture [T, S * G], and -- with the state posteriors `state_resps`
    [T, S] (None: ones) -- the accumulated statistics of the Gaussians [S * G, Q] and of the
    weights [S, G].  beer/models/mixtureset.py:85-112 (mixture.py:70-102 for S = 1) through
    oracle/beer_oracle.py, in `dtype` arithmetic (float32: the reference's own float32 run)."""
    from helpers import orc
    Xn = npy(X).astype(dtype) if hasattr(X, 'cpu') else np.asarray(X, dtype=dtype)
    p = ns.means_precisions.posterior
    post = [npy(getattr(p.params, n)).astype(dtype) for n in p._std_params_def]
    cat = weights_model.categorical if hasattr(weights_model, 'categorical') \
        else weights_model.categoricalset
    conc = npy(cat.weights.posterior.params.concentrations).astype(dtype).reshape(S, G)
    stats = orc.SUFFSTATS[cov](Xn)
    exp_T = orc.FAMILIES[cov]['exp'](*post)
    lw = orc.log_weights(conc[0])[None] if S == 1 else orc.log_weights_set(conc)
    ln, cr = orc.mixtureset_estep(stats, exp_T, Xn.shape[1], lw.astype(dtype))
    sr = np.ones((len(Xn), S), dtype=dtype) if state_resps is None else \
        (npy(state_resps) if hasattr(state_resps, 'cpu') else np.asarray(state_resps)).astype(dtype)
    wstats, acc = orc.mixtureset_accumulate(stats, cr, sr)
    return dict(ln=ln, resps=cr.reshape(len(Xn), S * G), acc=acc, wstats=wstats)
