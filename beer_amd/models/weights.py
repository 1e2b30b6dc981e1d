"""Weights of discrete choices: mixing weights of a mixture, of the mixtures of
HMM states (`CategoricalSet`), and phone weights of a phone loop with a
Dirichlet or a truncated stick-breaking (Dirichlet-process) prior, the latter
optionally under a Gamma hyper-prior on its concentration.

Interface of beer/models/categorical.py:39-209 and categoricalset.py:15-64:
class names, `create(...)` arguments, the parameter attributes pickled models
carry (`weights`, `stickbreaking`, `concentration`, `ordering`) and the callback
protocol of the stick-breaking models (statistics are re-shaped BEFORE the
update, the hyper-prior moves AFTER it: SURVEY.md appendix B, quirk Q10).

Numerics: E[ln pi] of the Dirichlet models is one kernel call
(`beer_dirichlet_log_weights`, memoised per parameter version); the stick
bookkeeping -- ordering the sticks by count, the tail sums, the running sum of
E[ln(1 - v)] -- is `beer_sb_transform_stats` / `beer_sb_log_weights`
(csrc/expfam.hip), not a chain of sort / flip / cumsum launches.
"""

import torch

from .. import _hip
from ..dists import Dirichlet, Gamma
from .basemodel import Model
from .modelset import ModelSet
from .parameters import ConjugateBayesianParameter

__all__ = ['Categorical', 'SBCategorical', 'SBCategoricalHyperPrior', 'CategoricalSet']


def _twin(dist_cls, *std_params):
    'Parameter whose prior and posterior start from the same standard parameters.'
    make = lambda: dist_cls.from_std_parameters(*[p.clone() for p in std_params])   # noqa: E731
    return ConjugateBayesianParameter(make(), make())


# ---- Dirichlet-distributed weights ---------------------------------------------------------

class _DirichletWeights:
    '''What `Categorical` and `CategoricalSet` share: one parameter `weights`
    holding the Dirichlet(s), statistics by the Dirichlet's likelihood function.'''

    @classmethod
    def create(cls, weights, prior_strength=1.):
        return cls(_twin(Dirichlet, weights.detach() * prior_strength))

    @property
    def mean(self):
        return self.weights.value()

    def mean_field_factorization(self):
        return [[self.weights]]

    def sufficient_statistics(self, data):
        return self.weights.likelihood_fn.sufficient_statistics(data)

    def expected_log_likelihood(self, stats):
        return self.weights.likelihood_fn(self.weights.natural_form(), stats)

    def log_weights(self):
        'E[ln pi] of every category (of every row for a set): one kernel call, memoised.'
        return self.weights.posterior.log_weights()


class Categorical(_DirichletWeights, Model):
    'One categorical distribution under a Dirichlet prior.'

    def __init__(self, weights):
        super().__init__()
        self.weights = weights

    def accumulate(self, stats):
        return {self.weights: stats.sum(dim=0)}


class CategoricalSet(_DirichletWeights, ModelSet):
    'S categorical distributions (the mixture weights of S states), one Dirichlet row each.'

    def __init__(self, weights):
        super().__init__()
        self.weights = weights

    def accumulate(self, stats, resps):
        return {self.weights: resps.t() @ stats}

    def accumulate_from_jointresps(self, jointresps_stats):
        return {self.weights: jointresps_stats.sum(dim=0)}

    def __len__(self):
        return len(self.weights)

    def __getitem__(self, key):
        rows = self.weights[key]
        return type(self)(rows) if isinstance(key, slice) else Categorical(rows)


# ---- stick-breaking weights ------------------------------------------------------------------

class SBCategorical(Model):
    '''Categorical whose weights are built from sticks, pi_i = v_i prod_{j before i}
    (1 - v_j), v_i ~ Beta stored as a two-column Dirichlet; the sticks are kept in
    order of decreasing count (`ordering[r]` = category of stick r).'''

    @classmethod
    def _sticks(cls, truncation, prior_strength):
        ab = torch.ones(truncation, 2)
        ab[:, 1] = prior_strength
        return _twin(Dirichlet, ab)

    @classmethod
    def create(cls, truncation, prior_strength=1.):
        return cls(cls._sticks(truncation, prior_strength))

    def __init__(self, stickbreaking):
        super().__init__()
        self.stickbreaking = stickbreaking
        conc = stickbreaking.posterior.params.concentrations
        # (any truncation, like the reference: up to 1024 sticks the kernels rank them in
        # LDS, beyond that on the global arrays -- csrc/expfam.hip)
        self.ordering = torch.arange(conc.shape[0], device=conc.device)
        stickbreaking.register_callback(self._transform_stats, notify_before_update=True)

    # -- kernels
    def _conc(self):
        return _hip.on_device(self.stickbreaking.posterior.params.concentrations)

    def _transform_stats(self):
        '''Before the update: re-order the sticks by the counts that just arrived and
        turn the counts [P] into the sticks' statistics [P, 2].'''
        counts = _hip.on_device(self.stickbreaking.stats)
        ordering = torch.empty(counts.shape[0], dtype=torch.int64, device=counts.device)
        pairs = torch.empty(counts.shape[0], 2, dtype=counts.dtype, device=counts.device)
        _hip.call('beer_sb_transform_stats', _hip.dtype_code(counts.dtype), counts.shape[0],
                  _hip.ptr(counts), _hip.ptr(ordering), _hip.ptr(pairs))
        home = self.stickbreaking.stats.device                 # where the model lives
        self.ordering = ordering.to(home)
        self.stickbreaking.stats = pairs.to(home)

    def _log_weights_and_tail(self):
        '(E[ln pi] [P] in the categories order, sum_i E[ln(1 - v_i)] as a 0-dim tensor).'
        conc = self._conc()
        order = _hip.on_device(self.ordering)
        log_w = torch.empty(conc.shape[0], dtype=conc.dtype, device=conc.device)
        tail = torch.empty((), dtype=conc.dtype, device=conc.device)
        _hip.call('beer_sb_log_weights', _hip.dtype_code(conc.dtype), conc.shape[0],
                  _hip.ptr(conc), _hip.ptr(order), _hip.ptr(log_w), _hip.ptr(tail))
        return log_w, tail

    # -- model protocol
    @property
    def reverse_ordering(self):
        'Stick of every category: the inverse permutation of `ordering`.'
        inverse = torch.empty_like(self.ordering)
        inverse[self.ordering] = torch.arange(len(self.ordering), device=self.ordering.device)
        return inverse

    @property
    def mean(self):
        'E[pi] under independent sticks: E[v_i] prod_{j before i} E[1 - v_j].'
        conc = self.stickbreaking.posterior.params.concentrations
        ab = conc[self.ordering.to(conc.device)]
        total = ab.sum(dim=-1) + torch.finfo(ab.dtype).eps
        stick, rest = ab[:, 0] / total, (ab[:, 1] / total).cumprod(dim=0)
        stick[1:] = stick[1:] * rest[:-1]
        return stick[self.reverse_ordering.to(stick.device)]

    def sufficient_statistics(self, data):
        return data

    def mean_field_factorization(self):
        return [[self.stickbreaking]]

    def log_weights(self):
        return self._log_weights_and_tail()[0]

    def expected_log_likelihood(self, stats):
        return stats @ self.log_weights().to(dtype=stats.dtype, device=stats.device)

    def accumulate(self, stats):
        return {self.stickbreaking: stats.sum(dim=0)}


class SBCategoricalHyperPrior(SBCategorical):
    '''Stick-breaking categorical whose concentration (the second Beta parameter of
    every stick's prior) has a Gamma prior: after every update of the sticks the
    Gamma posterior takes one full natural-gradient step on (sum_i E[ln(1 - v_i)], P)
    and writes its mean back into the sticks' prior.'''

    @classmethod
    def create(cls, truncation, prior_strength=1., hyper_prior_strength=1.):
        mean = torch.ones(1) * prior_strength
        shape = torch.ones_like(mean) * hyper_prior_strength
        concentration = _twin(Gamma, shape, hyper_prior_strength / mean)
        return cls(cls._sticks(truncation, prior_strength), concentration)

    def __init__(self, stickbreaking, concentration):
        super().__init__(stickbreaking)
        self.concentration = concentration
        stickbreaking.register_callback(self._on_stickbreaking_update)
        concentration.register_callback(self._on_concentration_update)
        self._on_concentration_update()

    def _on_concentration_update(self):
        self.stickbreaking.prior.params.concentrations[:, 1] = self.concentration.value()

    def _on_stickbreaking_update(self):
        _, tail = self._log_weights_and_tail()
        n_sticks = torch.full_like(tail, float(len(self.ordering)))
        self.concentration.stats = torch.stack([tail, n_sticks]).to(self.concentration.stats.device)
        self.concentration.natural_grad_update(lrate=1.)
