"""Model protocol: the drop-in boundary of the hot path.

API mirror of beer/models/basemodel.py:9-205 (same method names, arguments
and return conventions).
"""

import abc

import torch

from .parameters import ConjugateBayesianParameter

__all__ = ['Model', 'DiscreteLatentModel']


class Model(torch.nn.Module, metaclass=abc.ABCMeta):
    'Abstract base class of every model.'

    def __init__(self):
        super().__init__()
        self._cache = {}

    @property
    def cache(self):
        'Scratch results shared between expected_log_likelihood and accumulate.'
        return self._cache

    def clear_cache(self):
        self._cache = {}
        for module in self.modules():
            if module is not self and isinstance(module, Model):
                module._cache = {}

    def bayesian_parameters(self, paramtype=None, paramfilter=None, keepgroups=False):
        '''Iterate over the Bayesian parameters, in mean-field order.  With
        `keepgroups` yield one list per (non-empty) mean-field group.'''
        def keep(param):
            return (paramtype is None or type(param) == paramtype) and \
                   (paramfilter is None or paramfilter(param))

        for group in self.mean_field_factorization():
            selected = [param for param in group if keep(param)]
            if keepgroups:
                if selected:
                    yield selected
            else:
                yield from selected

    def conjugate_bayesian_parameters(self, keepgroups=False):
        return self.bayesian_parameters(paramtype=ConjugateBayesianParameter,
                                        keepgroups=keepgroups)

    def kl_div_posterior_prior(self):
        'KL(q || p) summed over every global parameter.'
        return sum([param.kl_div_posterior_prior().sum()
                    for param in self.bayesian_parameters()])

    def accumulated_statistics(self):
        return torch.cat([p.posterior.natural_parameters().reshape(-1)
                          - p.prior.natural_parameters().reshape(-1)
                          for p in self.bayesian_parameters()])

    @staticmethod
    def _swap_params(module, paramsmap):
        for name, child in module.named_children():
            if child in paramsmap:
                module.add_module(name, paramsmap[child])
            elif isinstance(child, Model):
                Model._swap_params(child, paramsmap)

    def replace_parameters(self, paramsmap):
        Model._swap_params(self, paramsmap)

    def __getstate__(self):
        state = self.__dict__.copy()
        state['_cache'] = {}
        state.pop('_idx_memo', None)             # device-side index tensors (PhoneLoop)
        return state

    # -- to be implemented by concrete models ---------------------------------
    @abc.abstractmethod
    def accumulate(self, s_stats, parent_msg=None):
        'dict {parameter: accumulated statistics}.'

    @abc.abstractmethod
    def expected_log_likelihood(self, s_stats, **kwargs):
        'Per-frame expected log-likelihood, tensor [n_frames].'

    @abc.abstractmethod
    def mean_field_factorization(self):
        'List of lists of Bayesian parameters.'

    @abc.abstractmethod
    def sufficient_statistics(self, data):
        'Sufficient statistics of `data` [n_frames, dim].'


class DiscreteLatentModel(Model, metaclass=abc.ABCMeta):
    'Model with a discrete latent variable over a set of components.'

    def __init__(self, modelset):
        super().__init__()
        self.modelset = modelset

    @abc.abstractmethod
    def posteriors(self, data, **kwargs):
        'p(latent | data), tensor [n_frames, n_components].'
