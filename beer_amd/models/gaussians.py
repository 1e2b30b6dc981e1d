"""Normal and NormalSet: Gaussians with conjugate priors on mean and
precision (full / diagonal / isotropic covariance).

API mirror of beer/models/normal.py:78-144 and beer/models/normalset.py:
84-135.  Default priors follow normal.py:29-63 and normalset.py:28-75.
"""

import torch

from .. import kernels
from ..dists import IsotropicNormalGamma, NormalGamma, NormalWishart
from .basemodel import Model
from .modelset import ModelSet
from .parameters import ConjugateBayesianParameter

__all__ = ['Normal', 'NormalSet']


class UnknownCovarianceType(Exception):
    pass


COV_TYPES = ('full', 'diagonal', 'isotropic')


def _as_full_cov(cov, dim, tensorconf):
    'Accept a scalar, a diagonal or a full covariance.'
    if cov.dim() == 1 and cov.shape[0] == 1:
        return cov * torch.eye(dim, **tensorconf)
    if cov.dim() == 1:
        return cov.diag()
    return cov


def _default_pair(mean, cov, cov_type, prior_strength, tensorconf, size=None, noise_std=0.):
    '''Prior/posterior pair of the default parameterisation.  `size` None
    builds a single Normal, otherwise a set whose posterior means are the
    prior mean plus `noise_std * sqrt(diag cov) * N(0, 1)` (one torch.randn
    draw of shape [size, dim], as the reference does).'''
    dim = mean.shape[-1]
    cov = _as_full_cov(cov, dim, tensorconf)
    ps = torch.tensor(float(prior_strength), **tensorconf)
    if size is None:
        prior_mean = post_mean = mean
        rep = lambda t: t                                   # noqa: E731
    else:
        prior_mean = mean.repeat(size, 1)
        noise = torch.randn(size, dim, **tensorconf) * noise_std * cov.diag().sqrt()[None, :]
        post_mean = prior_mean + noise
        rep = lambda t: t.repeat(size, 1)                   # noqa: E731
    if cov_type == 'full':
        dof = torch.tensor(float(prior_strength + dim - 1), **tensorconf)
        if size is None:
            rest = (ps, cov.inverse() / dof, dof)
        else:
            dofs = rep(dof)
            rest = (rep(ps), cov.inverse() / dofs[:, :, None], dofs)
        family = NormalWishart
    elif cov_type == 'diagonal':
        rates = prior_strength * cov.diag()
        rest = (rep(ps), rep(ps), rates if size is None else rates.repeat(size, 1))
        family = NormalGamma
    else:
        rate = prior_strength * cov.diag().max()
        rest = (rep(ps), rep(ps), rate if size is None else rate.repeat(size, 1))
        family = IsotropicNormalGamma
    prior = family.from_std_parameters(prior_mean, *rest)
    posterior = family.from_std_parameters(post_mean, *[t.clone() for t in rest])
    return ConjugateBayesianParameter(prior, posterior)


def _check(cov_type):
    if cov_type not in COV_TYPES:
        raise UnknownCovarianceType(f'Unknown covariance type: "{cov_type}"')


class Normal(Model):
    'Normal density with a prior over its mean and precision.'

    @classmethod
    def create(cls, mean, cov, prior_strength=1., cov_type='full'):
        _check(cov_type)
        tensorconf = {'dtype': mean.dtype, 'device': mean.device, 'requires_grad': False}
        return cls(_default_pair(mean.detach(), cov.detach(), cov_type, prior_strength,
                                 tensorconf))

    def __init__(self, mean_precision):
        super().__init__()
        self.mean_precision = mean_precision

    @property
    def mean(self):
        return self.mean_precision.value()[0]

    @property
    def cov(self):
        precision = self.mean_precision.value()[1]
        if precision.dim() == 2:
            return precision.inverse()
        if precision.dim() == 1 and precision.shape[0] > 1:
            return (1. / precision).diag()
        dim = len(self.mean)
        return (1. / precision) * torch.eye(dim, dtype=precision.dtype, device=precision.device)

    def sufficient_statistics(self, data):
        return self.mean_precision.likelihood_fn.sufficient_statistics(data)

    def mean_field_factorization(self):
        return [[self.mean_precision]]

    def expected_log_likelihood(self, stats):
        nparams = self.mean_precision.natural_form()
        return self.mean_precision.likelihood_fn(nparams, stats)

    def accumulate(self, stats, parent_msg=None):
        fn = self.mean_precision.likelihood_fn
        if kernels.is_dense(stats):
            acc = kernels.dense_accumulate(stats, None, None, 1, 1)
        else:
            acc = kernels.normal_accumulate(stats, None, None, 1, 1, fn.cov_type)
        ref = self.mean_precision.stats
        return {self.mean_precision: acc.view(-1).to(dtype=ref.dtype, device=ref.device)}


class NormalSet(ModelSet):
    'Set of Normal densities (the Gaussians of a mixture / of HMM states).'

    @classmethod
    def create(cls, mean, cov, size, prior_strength=1, noise_std=1., cov_type='full',
               shared_cov=False):
        if shared_cov:
            import warnings
            warnings.warn('The "NormalSet" with shared covariance is not supported '
                          'anymore. The argument will be ignored.', DeprecationWarning,
                          stacklevel=2)
        _check(cov_type)
        tensorconf = {'dtype': mean.dtype, 'device': mean.device, 'requires_grad': False}
        return cls(_default_pair(mean.detach(), cov.detach(), cov_type, prior_strength,
                                 tensorconf, size=size, noise_std=noise_std))

    def __init__(self, means_precisions):
        super().__init__()
        self.means_precisions = means_precisions

    @property
    def cov_type(self):
        return self.means_precisions.likelihood_fn.cov_type

    def sufficient_statistics(self, data):
        return self.means_precisions.likelihood_fn.sufficient_statistics(data)

    def mean_field_factorization(self):
        return [[self.means_precisions]]

    def expected_log_likelihood(self, stats):
        'Per-Gaussian expected log-likelihood [T, K] (fused kernel, no [T,Q]).'
        nparams = self.means_precisions.natural_form()
        return self.means_precisions.likelihood_fn(nparams, stats)

    def accumulate(self, stats, resps):
        'resps^T @ phi(X) -> [K, Q], accumulated in fp64 on the GPU.'
        K = len(self)
        if kernels.is_dense(stats):
            acc = kernels.dense_accumulate(stats, resps, None, K, 1)
        else:
            acc = kernels.normal_accumulate(stats, resps, None, K, 1, self.cov_type)
        ref = self.means_precisions.stats
        return {self.means_precisions: acc.to(dtype=ref.dtype, device=ref.device)}

    def __len__(self):
        return len(self.means_precisions)

    def __getitem__(self, key):
        if isinstance(key, slice):
            return self.__class__(self.means_precisions[key])
        return Normal(self.means_precisions[key])
