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
