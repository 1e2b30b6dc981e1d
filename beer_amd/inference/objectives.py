"""Evidence lower bound: the objective of variational-Bayes training.

API mirror of beer/inference/objectives.py (EvidenceLowerBoundInstance 54-116,
evidence_lower_bound 119-190).  Bookkeeping quirks kept on purpose
(SURVEY.md appendix B): the global KL term is subtracted once per call (Q1)
-- though it is only *computed* once per parameter version -- and `backward`
scales the summed statistics by datasize / sum(minibatch sizes) and stores
(overwrites) them in the parameters (Q2).
"""

import threading

import torch

from .. import _hip
from ..hmm_kernels import segment_sum

__all__ = ['evidence_lower_bound', 'EvidenceLowerBoundInstance']


def add_acc_stats(acc_stats1, acc_stats2):
    'Key-wise sum of two dictionaries of accumulated statistics.'
    out = dict(acc_stats1)
    for key, val in acc_stats2.items():
        out[key] = out[key] + val if key in out else val
    return out


def scale_acc_stats(acc_stats, scale):
    return {key: scale * val for key, val in acc_stats.items()}


class EvidenceLowerBoundInstance:
    '''ELBO of some data given a model, with the statistics needed for the
    update.  Created by `evidence_lower_bound`, summed with `+`.'''

    def __init__(self, value, acc_stats, model_parameters, minibatchsize, datasize):
        self.value = value
        self._acc_stats = acc_stats
        self._model_parameters = set(model_parameters)
        self._minibatchsize = minibatchsize
        self._datasize = datasize

    def __repr__(self):
        return f'EvidenceLowerBoundInstance(value={self.value})'

    def __float__(self):
        value = self.value
        return float(value.detach() if isinstance(value, torch.Tensor) else value)

    def __add__(self, other):
        if not isinstance(other, EvidenceLowerBoundInstance):
            raise ValueError('EvidenceLowerBoundInstance')
        if self._datasize != other._datasize:
            raise ValueError('Cannot add ELBOs evaluated on different data set')
        return EvidenceLowerBoundInstance(
            self.value + other.value,
            add_acc_stats(self._acc_stats, other._acc_stats),
            self._model_parameters.union(other._model_parameters),
            self._minibatchsize + other._minibatchsize,
            self._datasize)

    def backward(self, std_params=True):
        '''Hand the scaled statistics to the conjugate parameters (and
        back-propagate through non-conjugate ones when the value carries a
        graph).'''
        if std_params and isinstance(self.value, torch.Tensor) and self.value.requires_grad:
            (-self.value).backward()
        scale = self._datasize / self._minibatchsize
        for parameter in self._model_parameters:
            try:
                parameter.store_stats(scale * self._acc_stats[parameter])
            except KeyError:
                pass

    def sync(self, model):
        'Re-attach to the parameters of `model` (after unpickling).'
        self._model_parameters = set(model.bayesian_parameters())


_span_lock = threading.Lock()
_spans = {}                 # (device, n) -> int64 device tensor [0, n]: constants, made once


def _span(n, device):
    key = (device, n)
    with _span_lock:
        off = _spans.get(key)
        if off is None:
            if len(_spans) >= 64:
                _spans.clear()
            # (a host -> device copy: not while a HIP graph is being captured -- the warm-up
            # iteration in front of every capture has made the entry by then)
            off = _spans[key] = torch.tensor([0, n], dtype=torch.int64, device=device)
    return off


def frame_sum(values):
    'Sum of a per-frame tensor in fp64 on the device (beer_segment_sum).'
    values = _hip.on_device(values).reshape(-1)
    return segment_sum(values, _span(values.numel(), values.device), 1)[0]


def evidence_lower_bound(model=None, minibatch_data=None, datasize=-1, **kwargs):
    '''ELBO of `minibatch_data` given `model`, scaled to a data set of
    `datasize` frames.  With only `datasize` returns an empty accumulator.
    Extra keyword arguments go to `model.expected_log_likelihood`
    (`labels` for Mixture; `inference_graph`, `viterbi`, `state_path`,
    `scale` for HMM / PhoneLoop).'''
    if model is None and minibatch_data is None and datasize > 0:
        return EvidenceLowerBoundInstance(0., {}, [], 0, datasize)
    if model is None or minibatch_data is None:
        raise ValueError('if datasize is not provided, need at least "model" '
                         'and "minibatch_data"')
    mb_datasize = len(minibatch_data)
    if datasize <= 0:
        datasize = mb_datasize
    scale = datasize / float(mb_datasize)
    stats = model.sufficient_statistics(minibatch_data)
    exp_llh = model.expected_log_likelihood(stats, **kwargs)
    kl_div = torch.as_tensor(model.kl_div_posterior_prior())
    if isinstance(exp_llh, torch.Tensor) and exp_llh.requires_grad:
        # models with non-conjugate (torch.nn) parameters: VAE.  The sum stays
        # in the autograd graph so that `elbo.backward()` reaches them.
        total = _hip.on_device(exp_llh).to(torch.float64).sum()
    else:
        total = frame_sum(exp_llh)                              # fp64 device scalar
    elbo_value = float(scale) * total - kl_div.to(total.device, torch.float64)
    acc_stats = model.accumulate(stats)
    model.clear_cache()
    return EvidenceLowerBoundInstance(elbo_value, acc_stats, model.bayesian_parameters(),
                                      mb_datasize, datasize)
