"""Sets of models addressed by an index (the per-state emission densities of an
HMM): concatenation of sets, a view through per-call pdf ids, repetition.

Interface of beer/models/modelset.py:9-211 -- class names, constructor arguments
and the attributes pickled models carry (`modelsets`, `original_modelset`,
`modelset`, `repeat`).  The index arithmetic that the reference does with torch
one-liners on [T, S] tensors runs in the gather / scatter kernels of
csrc/hmm.hip (`hmm_kernels.gather_columns / scatter_columns`).
"""

import abc

import torch

from .basemodel import Model

__all__ = ['DynamicallyOrderedModelSet', 'JointModelSet', 'ModelSet', 'RepeatedModelSet']


class ModelSet(Model, metaclass=abc.ABCMeta):
    'A model that is a sequence of models: `len(set)` members, `set[i]` the i-th.'

    @abc.abstractmethod
    def __getitem__(self, key):
        pass

    @abc.abstractmethod
    def __len__(self):
        pass


class _Wrapper(ModelSet):
    '''A set defined on top of ONE inner set (attribute `_inner_name`): statistics
    and mean-field groups are the inner set's.'''
    _inner_name = None

    @property
    def _inner(self):
        return getattr(self, self._inner_name)

    def mean_field_factorization(self):
        return self._inner.mean_field_factorization()

    def sufficient_statistics(self, data):
        return self._inner.sufficient_statistics(data)


class JointModelSet(ModelSet):
    '''Several sets side by side (e.g. silence and speech emission groups with
    different numbers of components): member i of the joint set is member
    i - first of the set whose span [first, first + len) holds i.'''

    def __init__(self, modelsets):
        super().__init__()
        self.modelsets = torch.nn.ModuleList(modelsets)

    def _spans(self):
        'Yield (set, first index, one past its last index).'
        first = 0
        for member in self.modelsets:
            yield member, first, first + len(member)
            first += len(member)

    def mean_field_factorization(self):
        # every member must come as ONE group; the joint set is their union
        union = []
        for member in self.modelsets:
            groups = member.mean_field_factorization()
            if len(groups) > 1:
                raise ValueError('Invalid model set: more than 1 mean field group')
            union.extend(groups[0])
        return [union]

    def sufficient_statistics(self, data):
        # (all members share the statistics of the first: same family)
        return self.modelsets[0].sufficient_statistics(data)

    def expected_log_likelihood(self, stats):
        columns = [member.expected_log_likelihood(stats) for member in self.modelsets]
        return columns[0] if len(columns) == 1 else torch.cat(columns, dim=-1)

    def accumulate(self, stats, resps):
        collected = {}
        for member, first, last in self._spans():
            collected.update(member.accumulate(stats, resps[:, first:last]))
        return collected

    def __getitem__(self, key):
        if key < 0:
            raise ValueError('Unsupported negative index')
        for member, first, last in self._spans():
            if key < last:
                return member[key - first]
        raise IndexError('index out of range')

    def __len__(self):
        return sum(len(member) for member in self.modelsets)


class DynamicallyOrderedModelSet(_Wrapper):
    '''The inner set seen through a list of pdf ids given per call: column s of the
    log-likelihoods is the inner set's column order[s]; ids may repeat (states of
    an alignment graph that share a pdf), and their responsibilities then add up.'''
    _inner_name = 'original_modelset'

    def __init__(self, original_modelset):
        super().__init__()
        self.original_modelset = original_modelset

    def expected_log_likelihood(self, stats, order=None):
        from ..hmm_kernels import gather_columns
        self.cache['order'] = list(range(len(self._inner))) if order is None else order
        return gather_columns(self._inner.expected_log_likelihood(stats), self.cache['order'])

    def accumulate(self, stats, resps):
        from ..hmm_kernels import scatter_columns
        by_pdf = scatter_columns(resps, self.cache['order'], len(self._inner))
        return self._inner.accumulate(stats, by_pdf)

    def __getitem__(self, key):
        return self._inner[key]

    def __len__(self):
        return len(self._inner)


class RepeatedModelSet(_Wrapper):
    '''`repeat` copies of the inner set in a row (the same components offered to
    several classes): likelihood columns are tiled, responsibilities of the copies
    are summed back onto the one set of parameters.'''
    _inner_name = 'modelset'

    def __init__(self, modelset, repeat):
        super().__init__()
        self.modelset, self.repeat = modelset, repeat

    def expected_log_likelihood(self, stats):
        once = self._inner.expected_log_likelihood(stats)
        return once.repeat(1, self.repeat)

    def accumulate(self, stats, resps):
        per_copy = resps.reshape(resps.shape[0], self.repeat, len(self._inner))
        return self._inner.accumulate(stats, per_copy.sum(dim=1))

    def __getitem__(self, key):
        return self._inner[key % len(self._inner)]

    def __len__(self):
        return self.repeat * len(self._inner)
