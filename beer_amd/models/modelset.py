"""Containers of models: concatenation, dynamic re-ordering by pdf id,
repetition.  API mirror of beer/models/modelset.py:9-211."""

import abc

import torch

from .basemodel import Model

__all__ = ['DynamicallyOrderedModelSet', 'JointModelSet', 'ModelSet', 'RepeatedModelSet']


class ModelSet(Model, metaclass=abc.ABCMeta):
    'Iterable set of models sharing one type of sufficient statistics.'

    @abc.abstractmethod
    def __getitem__(self, key):
        pass

    @abc.abstractmethod
    def __len__(self):
        pass


class JointModelSet(ModelSet):
    'Concatenation of model sets (e.g. silence + speech emission groups).'

    def __init__(self, modelsets):
        super().__init__()
        self.modelsets = torch.nn.ModuleList(modelsets)

    def mean_field_factorization(self):
        merged = []
        for modelset in self.modelsets:
            groups = modelset.mean_field_factorization()
            if len(groups) > 1:
                raise ValueError('Invalid model set: more than 1 mean field group')
            merged += groups[0]
        return [merged]

    def sufficient_statistics(self, data):
        return self.modelsets[0].sufficient_statistics(data)

    def expected_log_likelihood(self, stats):
        return torch.cat([m.expected_log_likelihood(stats) for m in self.modelsets], dim=-1)

    def accumulate(self, stats, resps):
        acc, first = {}, 0
        for modelset in self.modelsets:
            n = len(modelset)
            acc.update(modelset.accumulate(stats, resps[:, first:first + n]))
            first += n
        return acc

    def __getitem__(self, key):
        if key < 0:
            raise ValueError('Unsupported negative index')
        first = 0
        for modelset in self.modelsets:
            if key < first + len(modelset):
                return modelset[key - first]
            first += len(modelset)
        raise IndexError('index out of range')

    def __len__(self):
        return sum(len(m) for m in self.modelsets)


class DynamicallyOrderedModelSet(ModelSet):
    '''View of a model set through a per-call list of pdf ids (ids may
    repeat: parameter sharing between states of an alignment graph).'''

    def __init__(self, original_modelset):
        super().__init__()
        self.original_modelset = original_modelset

    def mean_field_factorization(self):
        return self.original_modelset.mean_field_factorization()

    def sufficient_statistics(self, data):
        return self.original_modelset.sufficient_statistics(data)

    def expected_log_likelihood(self, stats, order=None):
        from ..hmm_kernels import gather_columns
        if order is None:
            order = list(range(len(self.original_modelset)))
        pc_exp_llh = self.original_modelset.expected_log_likelihood(stats)
        self.cache['order'] = order
        return gather_columns(pc_exp_llh, order)

    def accumulate(self, stats, resps):
        from ..hmm_kernels import scatter_columns
        order = self.cache['order']
        new_resps = scatter_columns(resps, order, len(self.original_modelset))
        return self.original_modelset.accumulate(stats, new_resps)

    def __getitem__(self, key):
        return self.original_modelset[key]

    def __len__(self):
        return len(self.original_modelset)


class RepeatedModelSet(ModelSet):
    'A model set repeated `repeat` times (components shared across classes).'

    def __init__(self, modelset, repeat):
        super().__init__()
        self.modelset = modelset
        self.repeat = repeat

    def mean_field_factorization(self):
        return self.modelset.mean_field_factorization()

    def sufficient_statistics(self, data):
        return self.modelset.sufficient_statistics(data)

    def expected_log_likelihood(self, stats):
        llhs = self.modelset.expected_log_likelihood(stats)
        return llhs[:, None, :].repeat(1, self.repeat, 1).view(len(stats), -1)

    def accumulate(self, stats, resps):
        new_resps = resps.reshape(len(stats), self.repeat, -1).sum(dim=1)
        return self.modelset.accumulate(stats, new_resps)

    def __getitem__(self, key):
        return self.modelset[key % len(self.modelset)]

    def __len__(self):
        return len(self.modelset) * self.repeat
