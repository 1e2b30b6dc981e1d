"""Concrete conjugate pairs: Normal-Wishart, Normal-Gamma, isotropic
Normal-Gamma, Dirichlet, Gamma -- and their conjugate likelihoods.

API mirror of beer/dists/{normalwishart,normalgamma,isonormalgamma,dirichlet,
gamma}.py.  The arithmetic lives in beer_amd/csrc/expfam.hip (fp64 internally
whatever the storage type); this file only shapes arguments.
"""

import math

import torch

from .. import _hip
from ..stats import FrameStats, reference_layout_enabled
from .expfam import ConjugateLikelihood, ExponentialFamily, make_std_params

__all__ = [
    'NormalLikelihood', 'NormalWishart', 'NormalWishartStdParams',
    'NormalDiagonalLikelihood', 'NormalGamma', 'NormalGammaStdParams',
    'IsotropicNormalLikelihood', 'IsotropicNormalGamma',
    'IsotropicNormalGammaStdParams',
    'CategoricalLikelihood', 'Dirichlet', 'DirichletStdParams',
    'GammaLikelihood', 'Gamma', 'GammaStdParams',
]


def _run(name, dtype, ints, ins, outs):
    'Launch `name(dtype, *ints, *ins, *outs, stream)` on device copies.'
    code = _hip.dtype_code(dtype)
    dins = [_hip.on_device(t, dtype) for t in ins]
    _hip.call(name, code, *ints, *[_hip.ptr(t) for t in dins], *[_hip.ptr(t) for t in outs])


def _empty(shape, like, dtype):
    dev = _hip.require_device() if like.device.type != 'cuda' else like.device
    return torch.empty(shape, dtype=dtype, device=dev)


# ---------------------------------------------------------------------------
# Gaussian families: shared driver
# ---------------------------------------------------------------------------

class _NormalFamily(ExponentialFamily):
    '''Common driver for the three mean/precision priors.  Subclasses give
    `_prefix` (kernel family), `_cov` and `_qdim(D)`.'''
    _std_params_def = {}
    _std_params_cls = None
    _prefix = None

    def __len__(self):
        shape = self.params.mean.shape
        return 1 if len(shape) <= 1 else shape[0]

    def natural_shape(self):
        'Shape of the natural parameters / accumulated statistics.'
        mean = self.params.mean
        Q = self._qdim(mean.shape[-1])
        return (Q,) if mean.dim() <= 1 else (mean.shape[0], Q)

    def _geometry(self):
        mean = self.params.mean
        single = mean.dim() <= 1
        return single, (1 if single else mean.shape[0]), mean.shape[-1]

    def _launch(self, which, width):
        mean = self.params.mean
        single, K, D = self._geometry()
        dtype = mean.dtype
        out = _empty((K, width) if width else (K,), mean, dtype)
        _run(f'beer_{self._prefix}_{which}', dtype, (K, D), self._tensors(), (out,))
        out = out.to(mean.device)
        if single and width:
            return out.view(-1)
        return out

    def expected_sufficient_statistics(self):
        return self._memoised('exp', lambda: self._launch(
            'expected_stats', self._qdim(self.params.mean.shape[-1])))

    def natural_parameters(self):
        return self._memoised('nat', lambda: self._launch(
            'natural', self._qdim(self.params.mean.shape[-1])))

    def log_norm(self):
        return self._memoised('lnorm', lambda: self._launch('log_norm', 0))


def _normal_from_natural(prefix, out_shapes, single_shapes):
    def from_natural(cls, natural_params):
        eta = natural_params
        single = eta.dim() == 1
        if single:
            eta = eta.view(1, -1)
        home, dtype = eta.device, eta.dtype
        K, Q = eta.shape
        D = cls._dim_from_q(Q)
        deta = _hip.on_device(eta)
        outs = [torch.empty(shape(K, D), dtype=dtype, device=deta.device)
                for shape in out_shapes]
        _hip.call(f'beer_{prefix}_from_natural', _hip.dtype_code(dtype), K, D,
                  _hip.ptr(deta), *[_hip.ptr(o) for o in outs])
        outs = [o.to(home) for o in outs]
        if single:
            outs = [o.view(*shape(D)) for o, shape in zip(outs, single_shapes)]
        return cls(*outs)
    return from_natural


# ---------------------------------------------------------------------------
# Normal-Wishart (full covariance)
# ---------------------------------------------------------------------------

class NormalLikelihood(ConjugateLikelihood):
    'Full-covariance Normal likelihood, statistics [x, -.5 vec(xx^T), -.5, .5].'
    cov_type = 'full'

    def __init__(self, dim):
        self.dim = dim

    def __eq__(self, other):
        return type(other) is type(self) and other.dim == self.dim

    def __repr__(self):
        return f'{type(self).__name__}(dim={self.dim})'

    def sufficient_statistics_dim(self, zero_stats=True):
        d = self.dim
        return 2 * d + d * (d - 1) // 2 + (2 if zero_stats else 0)

    @classmethod
    def sufficient_statistics(cls, data):
        '''Lazy statistics: the [T, Q] tensor of the reference
        (normalwishart.py:30-38) is only formed by `.dense()`.  Frames that
        carry an autograd graph (samples of a VAE's latent variable,
        vae.py:73) get the dense, differentiable tensor.'''
        if torch.is_grad_enabled() and data.requires_grad:
            from ..kernels import differentiable_stats
            return differentiable_stats(data, cls.cov_type)
        stats = FrameStats(data, cls.cov_type)
        return stats.dense() if reference_layout_enabled() else stats

    def __call__(self, pdfvecs, stats):
        'stats @ pdfvecs^T - D/2 ln 2pi -> [T, K] (normalwishart.py:88-92).'
        from ..kernels import dense_llh_autograd, is_dense, normal_llh_autograd
        if pdfvecs.dim() == 1:
            pdfvecs = pdfvecs.view(1, -1)
        if is_dense(stats):
            return dense_llh_autograd(stats, pdfvecs, self.dim)
        return normal_llh_autograd(stats, pdfvecs, self.cov_type)


class NormalDiagonalLikelihood(NormalLikelihood):
    'Diagonal-covariance Normal likelihood, statistics [x, -.5 x^2, -.5, .5].'
    cov_type = 'diagonal'

    def sufficient_statistics_dim(self, zero_stats=True):
        return 2 * self.dim + (2 if zero_stats else 0)


class IsotropicNormalLikelihood(NormalLikelihood):
    'Isotropic Normal likelihood, statistics [x, -.5 |x|^2, -.5, .5 D].'
    cov_type = 'isotropic'

    def sufficient_statistics_dim(self, zero_stats=True):
        return self.dim + 1 + (2 if zero_stats else 0)


NormalWishartStdParams = make_std_params(
    'NormalWishartStdParams', ('mean', 'scale', 'scale_matrix', 'dof'),
    _normal_from_natural(
        'nw',
        (lambda K, D: (K, D), lambda K, D: (K, 1), lambda K, D: (K, D, D), lambda K, D: (K, 1)),
        (lambda D: (D,), lambda D: (1,), lambda D: (D, D), lambda D: (1,))))
NormalWishartStdParams._dim_from_q = staticmethod(
    lambda Q: int(.5 * (-1 + math.sqrt(1 + 4 * (Q - 2)))))


class NormalWishart(_NormalFamily):
    _std_params_def = {
        'mean': 'Mean of the Normal pdf.',
        'scale': 'Scale of the precision of the Normal pdf.',
        'scale_matrix': 'Scale matrix of the Wishart pdf.',
        'dof': 'Degrees of freedom of the Wishart pdf.',
    }
    _std_params_cls = NormalWishartStdParams
    _prefix = 'nw'
    _cov = 'full'

    @staticmethod
    def _qdim(D):
        return D * D + D + 2

    @property
    def dim(self):
        d = self.params.mean.shape[-1]
        return (*self.params.mean.shape, (d, d))

    def conjugate(self):
        return NormalLikelihood(self.params.mean.shape[-1])

    def _stats_and_log_norm(self):
        '''E[T] and the log-normaliser from ONE factorisation of the scale matrices
        (a VB iteration asks for both); each lands in the memo of the other.'''
        mean = self.params.mean
        single, K, D = self._geometry()
        dtype = mean.dtype
        out = _empty((K, self._qdim(D)), mean, dtype)
        lnorm = _empty((K,), mean, dtype)
        _run('beer_nw_expected_stats_log_norm', dtype, (K, D), self._tensors(), (out, lnorm))
        out, lnorm = out.to(mean.device), lnorm.to(mean.device)
        return (out.view(-1) if single else out), lnorm

    def expected_sufficient_statistics(self):
        def both():
            exp, lnorm = self._stats_and_log_norm()
            self._memoised('lnorm', lambda: lnorm)
            return exp
        return self._memoised('exp', both)

    def log_norm(self):
        def both():
            exp, lnorm = self._stats_and_log_norm()
            self._memoised('exp', lambda: exp)
            return lnorm
        return self._memoised('lnorm', both)

    def update_from_natural_parameters(self, natural_params):
        '''The M-step in one launch (`beer_nw_update`): standard parameters from the
        natural ones and -- from the same factorisation -- E[T] and the log-normaliser
        of the new posterior, which the next iteration asks for (they go into the memo).'''
        eta = natural_params.detach()
        if eta.dim() != 2:
            return super().update_from_natural_parameters(natural_params)
        home, dtype = eta.device, eta.dtype
        K, Q = eta.shape
        D = self._std_params_cls._dim_from_q(Q)
        deta = _hip.on_device(eta)
        dev = deta.device
        new = lambda *shape: torch.empty(*shape, dtype=dtype, device=dev)          # noqa: E731
        mean, scale, W, dof = new(K, D), new(K, 1), new(K, D, D), new(K, 1)
        exp, lnorm = new(K, Q), new(K)
        _hip.call('beer_nw_update', _hip.dtype_code(dtype), K, D, _hip.ptr(deta), _hip.ptr(mean),
                  _hip.ptr(scale), _hip.ptr(W), _hip.ptr(dof), _hip.ptr(exp), _hip.ptr(lnorm),
                  None)
        self.params = self._std_params_cls(*[t.to(home) for t in (mean, scale, W, dof)])
        self.__dict__['_memo'] = {}
        exp, lnorm = exp.to(home), lnorm.to(home)
        self._memoised('nat', lambda: eta)
        self._memoised('exp', lambda: exp)
        self._memoised('lnorm', lambda: lnorm)

    def expected_value(self):
        'Expected mean and expected precision matrix (normalwishart.py:212-217).'
        if self.params.mean.dim() == 1:
            return self.params.mean, self.params.dof * self.params.scale_matrix
        return self.params.mean, self.params.dof[:, :, None] * self.params.scale_matrix


# ---------------------------------------------------------------------------
# Normal-Gamma (diagonal covariance)
# ---------------------------------------------------------------------------

NormalGammaStdParams = make_std_params(
    'NormalGammaStdParams', ('mean', 'scale', 'shape', 'rates'),
    _normal_from_natural(
        'ng',
        (lambda K, D: (K, D), lambda K, D: (K, 1), lambda K, D: (K, 1), lambda K, D: (K, D)),
        (lambda D: (D,), lambda D: (1,), lambda D: (1,), lambda D: (D,))))
NormalGammaStdParams._dim_from_q = staticmethod(lambda Q: (Q - 2) // 2)


class NormalGamma(_NormalFamily):
    _std_params_def = {
        'mean': 'Mean of the Normal.',
        'scale': 'Scale of the (diagonal) covariance matrix.',
        'shape': 'Shape parameter of the Gamma (shared across dimension).',
        'rates': 'Rate parameters of the Gamma.',
    }
    _std_params_cls = NormalGammaStdParams
    _prefix = 'ng'
    _cov = 'diagonal'

    @staticmethod
    def _qdim(D):
        return 2 * D + 2

    @property
    def dim(self):
        return (*self.params.mean.shape, self.params.rates.shape[-1])

    def conjugate(self):
        return NormalDiagonalLikelihood(self.params.mean.shape[-1])

    def expected_value(self):
        return self.params.mean, self.params.shape / self.params.rates


# ---------------------------------------------------------------------------
# Isotropic Normal-Gamma
# ---------------------------------------------------------------------------

IsotropicNormalGammaStdParams = make_std_params(
    'IsotropicNormalGammaStdParams', ('mean', 'scale', 'shape', 'rate'),
    _normal_from_natural(
        'ing',
        (lambda K, D: (K, D), lambda K, D: (K, 1), lambda K, D: (K, 1), lambda K, D: (K, 1)),
        (lambda D: (D,), lambda D: (1,), lambda D: (1,), lambda D: (1,))))
IsotropicNormalGammaStdParams._dim_from_q = staticmethod(lambda Q: Q - 3)


class IsotropicNormalGamma(_NormalFamily):
    _std_params_def = {
        'mean': 'Mean of the Normal.',
        'scale': 'Scale of the (isotropic) covariance matrix.',
        'shape': 'Shape parameter of the Gamma.',
        'rate': 'Rate parameter of the Gamma.',
    }
    _std_params_cls = IsotropicNormalGammaStdParams
    _prefix = 'ing'
    _cov = 'isotropic'

    @staticmethod
    def _qdim(D):
        return D + 3

    @property
    def dim(self):
        return (*self.params.mean.shape, 1)

    def conjugate(self):
        return IsotropicNormalLikelihood(self.params.mean.shape[-1])

    def expected_value(self):
        return self.params.mean, self.params.shape / self.params.rate


# ---------------------------------------------------------------------------
# Dirichlet / Categorical
# ---------------------------------------------------------------------------

class CategoricalLikelihood(ConjugateLikelihood):
    def __init__(self, dim):
        self.dim = dim

    def __eq__(self, other):
        return type(other) is type(self) and other.dim == self.dim

    def sufficient_statistics_dim(self, zero_stats=True):
        return self.dim - 1 + (1 if zero_stats else 0)

    def sufficient_statistics(self, data):
        '''Last column <- row sum (dirichlet.py:18-21).  Host-side glue on
        tiny [*, n_categories] tensors (phone / component counts).'''
        out = data.clone().reshape(-1, data.shape[-1])
        out[:, -1] = out.sum(dim=-1)
        return out.reshape(*data.shape)

    def __call__(self, pdfvecs, stats):
        return stats @ pdfvecs.t() if pdfvecs.dim() > 1 else stats @ pdfvecs


def _dirichlet_from_natural(cls, natural_params):
    eta = natural_params
    single = eta.dim() == 1
    eta2 = eta.view(1, -1) if single else eta
    home, dtype = eta.device, eta.dtype
    S, G = eta2.shape
    deta = _hip.on_device(eta2)
    out = torch.empty(S, G, dtype=dtype, device=deta.device)
    _hip.call('beer_dirichlet_from_natural', _hip.dtype_code(dtype), S, G,
              _hip.ptr(deta), _hip.ptr(out))
    out = out.to(home)
    return cls(out.view(-1) if single else out)


DirichletStdParams = make_std_params('DirichletStdParams', ('concentrations',),
                                     _dirichlet_from_natural)


class Dirichlet(ExponentialFamily):
    _std_params_def = {'concentrations': 'Concentrations parameter.'}
    _std_params_cls = DirichletStdParams

    def __len__(self):
        shape = self.params.concentrations.shape
        return 1 if len(shape) <= 1 else shape[0]

    def conjugate(self):
        return CategoricalLikelihood(self.params.concentrations.shape[-1])

    def natural_shape(self):
        return tuple(self.params.concentrations.shape)

    @property
    def dim(self):
        c = self.params.concentrations
        return len(c) if c.dim() <= 1 else tuple(c.shape)

    def _launch(self, name, reduce_rows=False):
        c = self.params.concentrations
        single = c.dim() <= 1
        c2 = c.reshape(1, -1) if single else c
        S, G = c2.shape
        out = _empty((S,) if reduce_rows else (S, G), c, c.dtype)
        _run(name, c.dtype, (S, G), (c2,), (out,))
        out = out.to(c.device)
        if reduce_rows:
            return out[0] if single else out
        return out.view(-1) if single else out

    def expected_sufficient_statistics(self):
        return self._memoised('exp', lambda: self._launch('beer_dirichlet_expected_stats'))

    def natural_parameters(self):
        return self._memoised('nat', lambda: self._launch('beer_dirichlet_natural'))

    def log_norm(self):
        return self._memoised('lnorm', lambda: self._launch('beer_dirichlet_log_norm', True))

    def log_weights(self):
        '''E[ln pi] for every category: the `eye -> sufficient_statistics ->
        stats @ E[T]` of Mixture._log_weights (mixture.py:45-48) in one call.'''
        return self._memoised('logw', lambda: self._launch('beer_dirichlet_log_weights'))

    def expected_value(self):
        c = self.params.concentrations
        return c / c.sum(dim=-1, keepdim=True)


# ---------------------------------------------------------------------------
# Gamma (hyper-prior on the stick-breaking concentration; 1 scalar in practice)
# ---------------------------------------------------------------------------

class GammaLikelihood(ConjugateLikelihood):
    def __init__(self, dim):
        self.dim = dim

    def sufficient_statistics_dim(self, zero_stats=True):
        return 2 * self.dim + (1 if zero_stats else 0)

    @staticmethod
    def sufficient_statistics(data):
        return torch.cat([-data, data.log(),
                          torch.ones(len(data), 1, dtype=data.dtype, device=data.device)],
                         dim=-1)

    def __call__(self, pdfvecs, stats):
        if pdfvecs.dim() == 1:
            pdfvecs = pdfvecs.view(1, -1)
        return stats @ pdfvecs.t()


def _gamma_from_natural(cls, natural_params):
    eta = natural_params.reshape(-1)
    home, dtype = eta.device, eta.dtype
    n = eta.shape[0] // 2
    deta = _hip.on_device(eta)
    shape = torch.empty(n, dtype=dtype, device=deta.device)
    rate = torch.empty(n, dtype=dtype, device=deta.device)
    _hip.call('beer_gamma_from_natural', _hip.dtype_code(dtype), n, _hip.ptr(deta),
              _hip.ptr(shape), _hip.ptr(rate))
    return cls(shape.to(home), rate.to(home))


GammaStdParams = make_std_params('GammaStdParams', ('shape', 'rate'), _gamma_from_natural)


class Gamma(ExponentialFamily):
    _std_params_def = {'shape': 'Shape parameter of the Gamma.',
                       'rate': 'Rate parameter of the Gamma.'}
    _std_params_cls = GammaStdParams

    def __len__(self):
        return 1

    @property
    def dim(self):
        return len(self.params.shape.reshape(-1))

    def conjugate(self):
        return GammaLikelihood(self.params.shape.shape[-1])

    def natural_shape(self):
        return (2 * self.params.shape.reshape(-1).shape[0],)

    def _launch(self, name, width):
        a = self.params.shape.reshape(-1)
        b = self.params.rate.reshape(-1)
        n = a.shape[0]
        out = _empty((width * n if width else 1,), a, a.dtype)
        _run(name, a.dtype, (n,), (a, b), (out,))
        out = out.to(a.device)
        return out if width else out[0]

    def expected_sufficient_statistics(self):
        return self._memoised('exp', lambda: self._launch('beer_gamma_expected_stats', 2))

    def natural_parameters(self):
        return self._memoised('nat', lambda: self._launch('beer_gamma_natural', 2))

    def log_norm(self):
        return self._memoised('lnorm', lambda: self._launch('beer_gamma_log_norm', 0))

    def expected_value(self):
        return self.params.shape / self.params.rate
