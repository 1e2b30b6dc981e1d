"""Categorical models of mixing / phone weights: Dirichlet prior, truncated
stick-breaking prior (with optional Gamma hyper-prior), and sets of
categoricals (the mixture weights of HMM states).

API mirror of beer/models/categorical.py:39-209 and
beer/models/categoricalset.py:15-64.  These carry at most a few hundred
numbers (one per phone / per component); besides the Dirichlet / Gamma kernel
calls the bookkeeping (sorting sticks, cumulative sums) is host-side glue, as
SURVEY.md section 8 row a15 prescribes.
"""

import torch

from ..dists import Dirichlet, Gamma
from .basemodel import Model
from .modelset import ModelSet
from .parameters import ConjugateBayesianParameter

__all__ = ['Categorical', 'SBCategorical', 'SBCategoricalHyperPrior', 'CategoricalSet']


def _dirichlet_param(weights, prior_strength):
    prior = Dirichlet.from_std_parameters(weights * prior_strength)
    posterior = Dirichlet.from_std_parameters(weights * prior_strength)
    return ConjugateBayesianParameter(prior, posterior)


def _stick_param(truncation, prior_strength):
    params = torch.ones(truncation, 2)
    params[:, 1] = prior_strength
    return ConjugateBayesianParameter(Dirichlet.from_std_parameters(params),
                                      Dirichlet.from_std_parameters(params.clone()))


def _concentration_param(mean, prior_strength):
    shape = torch.ones_like(mean) * prior_strength
    rate = prior_strength / mean
    return ConjugateBayesianParameter(Gamma.from_std_parameters(shape, rate),
                                      Gamma.from_std_parameters(shape.clone(), rate.clone()))


class Categorical(Model):
    'Categorical distribution with a Dirichlet prior.'

    @classmethod
    def create(cls, weights, prior_strength=1.):
        return cls(_dirichlet_param(weights.detach(), prior_strength))

    def __init__(self, weights):
        super().__init__()
        self.weights = weights

    @property
    def mean(self):
        return self.weights.value()

    def sufficient_statistics(self, data):
        return self.weights.likelihood_fn.sufficient_statistics(data)

    def mean_field_factorization(self):
        return [[self.weights]]

    def expected_log_likelihood(self, stats):
        return self.weights.likelihood_fn(self.weights.natural_form(), stats)

    def log_weights(self):
        'E[ln pi_k] for every category (one kernel call, memoised).'
        return self.weights.posterior.log_weights()

    def accumulate(self, stats):
        return {self.weights: stats.sum(dim=0)}


class SBCategorical(Model):
    'Categorical with a truncated stick-breaking (Dirichlet process) prior.'

    @classmethod
    def create(cls, truncation, prior_strength=1.):
        return cls(_stick_param(truncation, prior_strength))

    def __init__(self, stickbreaking):
        super().__init__()
        self.stickbreaking = stickbreaking
        device = stickbreaking.posterior.params.concentrations.device
        self.ordering = torch.arange(stickbreaking.posterior.dim[0], device=device)
        self.stickbreaking.register_callback(self._transform_stats,
                                             notify_before_update=True)

    def _transform_stats(self):
        '''Before the update: order the sticks by decreasing count and turn the
        counts [P] into stick statistics [P, 2] = (count_i, sum_{j>i} count_j)
        with the Dirichlet "last column <- row sum" convention.'''
        stats = self.stickbreaking.stats
        self.ordering = stats.sort(descending=True)[1]
        stats = stats[self.ordering]
        tail = torch.zeros_like(stats)
        tail[:-1] = stats[1:]
        tail = torch.flip(torch.flip(tail, dims=(0,)).cumsum(dim=0), dims=(0,))
        new_stats = torch.stack([stats, tail + stats], dim=-1)
        self.stickbreaking.stats = new_stats[self.reverse_ordering, :]

    def _log_v(self):
        c = self.stickbreaking.posterior.params.concentrations[self.ordering]
        s_dig = torch.digamma(c.sum(dim=-1))
        return torch.digamma(c[:, 0]) - s_dig, torch.digamma(c[:, 1]) - s_dig

    def _log_prob(self):
        log_v, log_1_v = self._log_v()
        log_prob = log_v
        log_prob[1:] += log_1_v[:-1].cumsum(dim=0)
        return log_prob, log_1_v

    @property
    def reverse_ordering(self):
        rev = torch.zeros_like(self.ordering)
        rev[self.ordering] = torch.arange(len(self.ordering), device=self.ordering.device)
        return rev

    @property
    def mean(self):
        c = self.stickbreaking.posterior.params.concentrations[self.ordering]
        norm = c.sum(dim=-1) + torch.finfo(c.dtype).eps
        weights = c[:, 0] / norm
        residual = (c[:, 1] / norm).cumprod(dim=0)
        weights[1:] *= residual[:-1]
        return weights[self.reverse_ordering]

    def sufficient_statistics(self, data):
        return data

    def mean_field_factorization(self):
        return [[self.stickbreaking]]

    def expected_log_likelihood(self, stats):
        return stats @ self.log_weights()

    def log_weights(self):
        log_prob, _ = self._log_prob()
        return log_prob[self.reverse_ordering]

    def accumulate(self, stats):
        return {self.stickbreaking: stats.sum(dim=0)}


class SBCategoricalHyperPrior(SBCategorical):
    'Stick-breaking categorical with a Gamma hyper-prior on the concentration.'

    @classmethod
    def create(cls, truncation, prior_strength=1., hyper_prior_strength=1.):
        concentration = _concentration_param(torch.ones(1) * prior_strength,
                                             hyper_prior_strength)
        return cls(_stick_param(truncation, prior_strength), concentration)

    def __init__(self, stickbreaking, concentration):
        super().__init__(stickbreaking)
        self.concentration = concentration
        self.stickbreaking.register_callback(self._on_stickbreaking_update)
        self.concentration.register_callback(self._on_concentration_update)
        self._on_concentration_update()

    def _on_concentration_update(self):
        self.stickbreaking.prior.params.concentrations[:, 1] = self.concentration.value()

    def _on_stickbreaking_update(self):
        _, log_1_v = self._log_prob()
        self.concentration.stats = torch.stack([log_1_v.sum(),
                                                log_1_v.new_tensor(float(len(log_1_v)))])
        self.concentration.natural_grad_update(lrate=1.)


class CategoricalSet(ModelSet):
    'Set of categoricals with Dirichlet priors (mixture weights of S states).'

    @classmethod
    def create(cls, weights, prior_strength=1.):
        return cls(_dirichlet_param(weights.detach(), prior_strength))

    def __init__(self, weights):
        super().__init__()
        self.weights = weights

    @property
    def mean(self):
        return self.weights.value()

    def sufficient_statistics(self, data):
        return self.weights.likelihood_fn.sufficient_statistics(data)

    def mean_field_factorization(self):
        return [[self.weights]]

    def expected_log_likelihood(self, stats):
        return self.weights.likelihood_fn(self.weights.natural_form(), stats)

    def log_weights(self):
        'E[ln pi_{s,g}] [S, G] (one kernel call, memoised).'
        return self.weights.posterior.log_weights()

    def accumulate(self, stats, resps):
        return {self.weights: resps.t() @ stats}

    def accumulate_from_jointresps(self, jointresps_stats):
        return {self.weights: jointresps_stats.sum(dim=0)}

    def __len__(self):
        return len(self.weights)

    def __getitem__(self, key):
        if isinstance(key, slice):
            return self.__class__(self.weights[key])
        return Categorical(self.weights[key])
