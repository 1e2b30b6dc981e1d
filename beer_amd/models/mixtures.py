"""Mixture and MixtureSet: the E-step of a GMM (config C1/C2) and of the GMM
emissions of HMM states (config C3).

API mirror of beer/models/mixture.py:14-115 and beer/models/mixtureset.py:
22-133.  `expected_log_likelihood` runs `beer_mixtureset_estep` (per-component
llh + logsumexp + responsibilities straight from the frames) and `accumulate`
runs `beer_normal_accumulate` (responsibility-weighted N_k, sum r x,
sum r xx^T in fp64): two kernel calls replace the reference's
cat / mul / mm / logsumexp / exp / mm sequence and its [T, Q] tensor.
"""

import torch

from .. import kernels
from .basemodel import DiscreteLatentModel
from .gaussians import NormalSet
from .modelset import ModelSet
from .weights import Categorical, CategoricalSet, SBCategorical

__all__ = ['Mixture', 'MixtureSet']


def _merge_groups(l1, l2):
    'Zip two mean-field factorizations, padding the shorter one.'
    l1, l2 = list(l1), list(l2)
    n = max(len(l1), len(l2))
    l1 += [[] for _ in range(n - len(l1))]
    l2 += [[] for _ in range(n - len(l2))]
    return [u + v for u, v in zip(l1, l2)]


def _fused(modelset):
    'True when the component set is a plain NormalSet the kernels can take.'
    return isinstance(modelset, NormalSet)


def _like(param, t):
    ref = param.stats
    return t.to(dtype=ref.dtype, device=ref.device)


class Mixture(DiscreteLatentModel):
    'Bayesian mixture model.'

    @classmethod
    def create(cls, modelset, categorical=None, prior_strength=1.):
        tensor = modelset.mean_field_factorization()[0][0].prior._tensors()[0]
        if categorical is None:
            weights = torch.ones(len(modelset), dtype=tensor.dtype, device=tensor.device)
            weights /= len(modelset)
            categorical = Categorical.create(weights, prior_strength)
        return cls(categorical, modelset)

    def __init__(self, categorical, modelset):
        super().__init__(modelset)
        self.categorical = categorical

    def _log_weights(self, tensorconf=None):
        return self.categorical.log_weights()

    def mean_field_factorization(self):
        return _merge_groups(self.modelset.mean_field_factorization(),
                             self.categorical.mean_field_factorization())

    def sufficient_statistics(self, data):
        return self.modelset.sufficient_statistics(data)

    def expected_log_likelihood(self, stats, labels=None, **kwargs):
        '''Per-frame ELBO term sum_k r ln N_k - sum_k r (ln r - E ln pi_k),
        i.e. logsumexp_k(l_tk + E ln pi_k); with `labels` the log-likelihood
        of the labelled component (mixture.py:70-93).'''
        if not _fused(self.modelset):
            raise NotImplementedError('Mixture components must be a NormalSet')
        ns = self.modelset
        K = len(ns)
        if kernels.is_dense(stats):
            return self._dense_expected_log_likelihood(stats, labels)
        wide = kernels.wide_mixture_split(stats, K, ns.cov_type) if labels is None else None
        if wide:
            # more than 256 components: blocks on the matrix cores, two-level softmax;
            # the cache holds the factored responsibilities (`.dense()` -> [T, K])
            log_norm, resps = kernels.wide_mixture_estep(
                stats, ns.means_precisions.natural_form(), self._log_weights().view(1, K), K,
                ns.cov_type, wide)
        else:
            log_norm, resps = kernels.mixtureset_estep(
                stats, ns.means_precisions.natural_form(), self._log_weights().view(1, K),
                1, K, ns.cov_type, labels=labels)
        self.cache['resps'] = resps
        value = log_norm.view(-1)
        if kernels.has_source(stats):
            # differentiable frames (one sample per frame of a VAE): the gradient flows
            # through sum_k r_k l_k only, as in the statistics-in variant below
            value = kernels.attach_frame_grad(
                stats, value, resps.dense() if wide else resps,
                ns.means_precisions.natural_form())
        return value

    def _dense_expected_log_likelihood(self, stats, labels):
        '''Statistics-in variant (prior of a VAE): same value, and the gradient
        w.r.t. the statistics flows through sum_k r_k l_k only (mixture.py:92).'''
        ns = self.modelset
        K = len(ns)
        fn = ns.means_precisions.likelihood_fn
        nparams = ns.means_precisions.natural_form()
        pc = kernels.dense_llh(stats, nparams, fn.dim)
        if labels is None:
            log_norm, resps = kernels.dense_softmax(pc, self._log_weights().view(1, K), 1, K)
            value = log_norm.view(-1)
        else:
            lab = torch.as_tensor(labels).to(device=pc.device, dtype=torch.int64).view(-1, 1)
            resps = torch.zeros_like(pc).scatter_(1, lab, 1.)
            value = kernels.rowdot(pc, resps)
        self.cache['resps'] = resps
        return kernels.attach_stats_grad(stats, value, resps, nparams)

    def accumulate(self, stats):
        ns = self.modelset
        K = len(ns)
        if kernels.is_dense(stats):
            acc = kernels.dense_accumulate(stats, self.cache['resps'], None, K, 1)
        else:
            acc = kernels.normal_accumulate(stats, self.cache['resps'], None, K, 1,
                                            ns.cov_type)
        wparam = self.categorical.mean_field_factorization()[0][0]
        if isinstance(self.categorical, SBCategorical):
            # stick-breaking weights take the raw counts N_k (categorical.py:149-151)
            wacc = -2. * acc[:, -2]
        else:
            wacc = kernels.weights_from_acc(acc, 1, K).view(-1)
        return {wparam: _like(wparam, wacc),
                ns.means_precisions: _like(ns.means_precisions, acc)}

    def posteriors(self, data):
        stats = self.sufficient_statistics(data)
        ns = self.modelset
        K = len(ns)
        _, resps = kernels.mixtureset_estep(
            stats, ns.means_precisions.natural_form(), self._log_weights().view(1, K),
            1, K, ns.cov_type)
        return resps


class MixtureSet(ModelSet):
    'Set of S mixtures with G components each (component k belongs to k // G).'

    @classmethod
    def create(cls, size, modelset, prior_strength=1.):
        n_comp = len(modelset) // size
        weights = torch.ones(size, n_comp) / n_comp
        return cls(CategoricalSet.create(weights, prior_strength), modelset)

    def __init__(self, categoricalset, modelset):
        super().__init__()
        self.categoricalset = categoricalset
        self.modelset = modelset

    @property
    def n_comp_per_mixture(self):
        return len(self.modelset) // len(self)

    def _log_weights(self, tensorconf=None):
        return self.categoricalset.log_weights()

    def mean_field_factorization(self):
        return _merge_groups(self.modelset.mean_field_factorization(),
                             self.categoricalset.mean_field_factorization())

    def sufficient_statistics(self, data):
        return self.modelset.sufficient_statistics(data)

    def expected_log_likelihood(self, stats):
        'Per-state mixture log-normaliser [T, S]; caches the component resps.'
        if not _fused(self.modelset):
            raise NotImplementedError('MixtureSet components must be a NormalSet')
        ns = self.modelset
        S, G = len(self), self.n_comp_per_mixture
        if kernels.is_dense(stats):
            # statistics-in: no gradient, the log-normaliser is detached
            # (mixtureset.py:93)
            fn = ns.means_precisions.likelihood_fn
            pc = kernels.dense_llh(stats, ns.means_precisions.natural_form(), fn.dim)
            log_norm, resps = kernels.dense_softmax(pc, self._log_weights(), S, G)
        else:
            log_norm, resps = kernels.mixtureset_estep(
                stats, ns.means_precisions.natural_form(), self._log_weights(), S, G,
                ns.cov_type)
        self.cache['resps'] = resps.view(-1, S, G)
        return log_norm

    def accumulate(self, stats, resps):
        'Joint (state x component) responsibilities -> weights + Gaussian stats.'
        ns = self.modelset
        S, G = len(self), self.n_comp_per_mixture
        comp = self.cache['resps'].reshape(-1, S * G)
        if kernels.is_dense(stats):
            acc = kernels.dense_accumulate(stats, comp, resps, S, G)
        else:
            acc = kernels.normal_accumulate(stats, comp, resps, S, G, ns.cov_type)
        wacc = kernels.weights_from_acc(acc, S, G)
        wparam = self.categoricalset.weights
        return {wparam: _like(wparam, wacc),
                ns.means_precisions: _like(ns.means_precisions, acc)}

    def __len__(self):
        return len(self.categoricalset)

    def __getitem__(self, key):
        ncpm = self.n_comp_per_mixture
        if isinstance(key, int):
            return Mixture(self.categoricalset[key],
                           self.modelset[slice(key * ncpm, (key + 1) * ncpm)])
        if isinstance(key, slice):
            start = 0 if key.start is None else key.start * ncpm
            stop = len(self) if key.stop is None else key.stop * ncpm
            step = 1 if key.step is None else key.step * ncpm
            return self.__class__(self.categoricalset[key],
                                  self.modelset[slice(start, stop, step)])
        raise IndexError(f'Unsupported index: {key}')
