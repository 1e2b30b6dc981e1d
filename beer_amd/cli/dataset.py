"""On-disk data set: an `.npz` archive of per-utterance [T, D] feature
matrices plus global mean / variance / frame count.  Field names and behaviour
follow beer/cli/dataset.py:10-80 so that pickles written by either tool load
in the other (see `compat.load`)."""

import random
from typing import NamedTuple

import numpy as np
import torch

__all__ = ['Dataset', 'Utterance']


class Utterance(NamedTuple):
    'An utterance id and its features (always float32, like the reference).'
    id: str
    features: torch.Tensor


class Dataset:
    'Collection of utterances backed by a features archive.'

    def __init__(self, feapath, mean, var, size, _fea_dict=None):
        self.feapath = feapath
        self.mean = mean
        self.var = var
        self.size = size
        self._fea_dict = _fea_dict

    @property
    def fea_dict(self):
        if self._fea_dict is None:
            self._fea_dict = np.load(self.feapath)
        return self._fea_dict

    def __getstate__(self):
        state = dict(self.__dict__)
        state['_fea_dict'] = None
        return state

    def __len__(self):
        return len(self.fea_dict.files)

    def utterances(self, random_order=True):
        'Iterate over the utterances (sorted by id unless `random_order`).'
        ids = sorted(self.fea_dict.keys())
        if random_order:
            random.shuffle(ids)
        for uttid in ids:
            yield self[uttid]

    def __getitem__(self, key):
        return Utterance(key, torch.from_numpy(self.fea_dict[key]).float())

    @classmethod
    def from_archive(cls, path):
        '''Global statistics of an archive (`beer dataset create`,
        cli/subcommands/dataset/create.py:14-40).'''
        feats = np.load(path)
        keys = list(feats.keys())
        dim = feats[keys[0]].shape[1]
        tot, tot2, count = np.zeros(dim), np.zeros(dim), 0
        for k in keys:
            x = feats[k]
            tot += x.sum(axis=0)
            tot2 += (x ** 2).sum(axis=0)
            count += len(x)
        mean = tot / count
        var = tot2 / count - mean ** 2
        return cls(path, torch.from_numpy(mean).float(), torch.from_numpy(var).float(),
                   int(count))
