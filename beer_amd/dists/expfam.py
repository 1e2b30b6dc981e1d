"""Exponential-family machinery shared by every prior/posterior pdf.

API mirror of beer/dists/basedist.py (ExponentialFamily 64-176,
ConjugateLikelihood 179-240, kl_div 243-263, ParametersView 39-61) -- same
class and method names, same argument meaning -- but every numerical method
is a call into the HIP library (beer_amd/csrc/expfam.hip) and results are
cached per parameter version instead of being recomputed on every E-step.
"""

import abc

import torch

from .. import _hip

__all__ = ['ConjugateLikelihood', 'ExponentialFamily', 'ParametersView',
           'kl_div', 'DistributionTypeMismatch', 'SupportDimensionMismatch']


class MissingParameterAttribute(Exception):
    pass


class UndefinedParameters(Exception):
    pass


class UndefinedStdParametersClass(Exception):
    pass


class DistributionTypeMismatch(Exception):
    'KL divergence between pdfs of different families.'


class SupportDimensionMismatch(Exception):
    'KL divergence between pdfs with different support.'


def make_std_params(clsname, fields, from_natural, module='beer_amd.dists.families'):
    '''Build the `<Family>StdParams` nn.Module class: one registered buffer per
    standard parameter + the `from_natural_parameters` classmethod.'''

    def __init__(self, *args, **kwargs):
        torch.nn.Module.__init__(self)
        values = dict(zip(fields, args))
        values.update(kwargs)
        if set(values) != set(fields):
            raise TypeError(f'{clsname} expects {fields}')
        for name in fields:
            self.register_buffer(name, values[name])

    def __repr__(self):
        body = ', '.join(f'{n}={getattr(self, n)}' for n in fields)
        return f'{clsname}({body})'

    return type(clsname, (torch.nn.Module,), {
        '__init__': __init__, '__repr__': __repr__, '_fields': tuple(fields),
        '__module__': module, '__qualname__': clsname,
        'from_natural_parameters': classmethod(from_natural),
    })


class ParametersView(torch.nn.Module):
    'Read-only slice `idx` of another parameter object (no own storage).'

    def __init__(self, ref, names, idx):
        super().__init__()
        object.__setattr__(self, '_ref', ref)
        self.idx = idx
        self.names = tuple(names)

    @property
    def ref(self):
        return self._ref

    def __getattr__(self, name):
        if name in ('names', 'idx', '_ref'):
            return super().__getattr__(name)
        if name in self.names:
            return getattr(self._ref, name)[self.idx]
        return super().__getattr__(name)

    @property
    def _fields(self):
        return self.names

    def from_natural_parameters(self, natural_parameters):
        return self._ref.from_natural_parameters(natural_parameters)

    def __repr__(self):
        body = ', '.join(f'{n}={getattr(self, n)}' for n in self.names)
        return f'view<{self._ref.__class__.__qualname__}({body})>'


class ExponentialFamily(torch.nn.Module, metaclass=abc.ABCMeta):
    '''A set of K same-family pdfs (K = 1 when the parameters carry no leading
    set dimension).  Subclasses declare `_std_params_def` (name -> doc) and
    `_std_params_cls`, exactly as in the reference.'''

    def __init_subclass__(cls):
        if not hasattr(cls, '_std_params_def'):
            raise UndefinedParameters('"_std_params_def" is missing')
        if not hasattr(cls, '_std_params_cls'):
            raise UndefinedStdParametersClass('"_std_params_cls" is missing')

    def __init__(self, params):
        super().__init__()
        self.params = params
        object.__setattr__(self, '_memo', {})

    @classmethod
    def from_std_parameters(cls, *args, **kwargs):
        return cls(cls._std_params_cls(*args, **kwargs))

    # -- per-version memoisation ------------------------------------------
    def _tensors(self):
        return tuple(getattr(self.params, n) for n in self._std_params_def)

    def _memoised(self, key, compute):
        '''The reference recomputes E[T], A(eta) ... on every call (K Cholesky
        per utterance).  Here a result stays valid until a
        parameter tensor is replaced or modified in place.'''
        memo = self.__dict__.setdefault('_memo', {})
        tensors = self._tensors()
        sig = tuple(t._version for t in tensors)
        hit = memo.get(key)
        if hit is not None and len(hit[0]) == len(tensors) and \
                all(a is b for a, b in zip(hit[0], tensors)) and hit[1] == sig:
            return hit[2]
        value = compute()
        memo[key] = (tensors, sig, value)
        return value

    def __getstate__(self):
        state = self.__dict__.copy()
        state['_memo'] = {}
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.__dict__['_memo'] = {}

    # -- subclass interface -------------------------------------------------
    @abc.abstractmethod
    def __len__(self):
        pass

    def __getitem__(self, idx):
        names = tuple(self._std_params_def.keys())
        return self.__class__(ParametersView(self.params, names, idx))

    @abc.abstractmethod
    def conjugate(self):
        'Conjugate likelihood descriptor.'

    @property
    @abc.abstractmethod
    def dim(self):
        'Dimension of the support.'

    @abc.abstractmethod
    def expected_sufficient_statistics(self):
        pass

    @abc.abstractmethod
    def expected_value(self):
        pass

    @abc.abstractmethod
    def log_norm(self):
        pass

    @abc.abstractmethod
    def natural_parameters(self):
        pass

    def sample(self, nsamples):
        raise NotImplementedError

    def update_from_natural_parameters(self, natural_params):
        self.params = self.params.from_natural_parameters(natural_params)
        self.__dict__['_memo'] = {}
        # eta <-> standard parameters is a bijection: what was just set IS the
        # natural form of the new parameters (the next natural-gradient step
        # would otherwise invert every scale matrix again to recover it)
        eta = natural_params.detach()
        self._memoised('nat', lambda: eta)


class ConjugateLikelihood(metaclass=abc.ABCMeta):
    'Likelihood function conjugate to an ExponentialFamily prior.'

    @abc.abstractmethod
    def sufficient_statistics_dim(self, zero_stats=True):
        pass

    @abc.abstractmethod
    def sufficient_statistics(self, data):
        pass

    @abc.abstractmethod
    def __call__(self, natural_parameters, stats):
        pass

    def pdfvectors_from_rvectors(self, rvecs):
        raise NotImplementedError('subspace (GSM) models are out of scope')

    def parameters_from_pdfvector(self, pdfvec):
        raise NotImplementedError('subspace (GSM) models are out of scope')


def kl_div(pdf1, pdf2):
    '''KL(pdf1 || pdf2), one value per pdf of the set
    (beer/dists/basedist.py:243-263), computed by `beer_kl_div`.'''
    if pdf1.__class__ is not pdf2.__class__:
        raise DistributionTypeMismatch(
            f'({pdf1.__class__} != {pdf2.__class__})')
    if pdf1.dim != pdf2.dim:
        raise SupportDimensionMismatch(f'({pdf1.dim} != {pdf2.dim})')
    eta1 = pdf1.natural_parameters()
    home = eta1.device
    single = eta1.dim() == 1
    dtype = eta1.dtype
    es = _hip.on_device(pdf1.expected_sufficient_statistics(), dtype)
    e1 = _hip.on_device(eta1)
    e2 = _hip.on_device(pdf2.natural_parameters(), dtype)
    l1 = _hip.on_device(pdf1.log_norm(), dtype).reshape(-1)
    l2 = _hip.on_device(pdf2.log_norm(), dtype).reshape(-1)
    Q = e1.shape[-1]
    K = 1 if single else e1.shape[0]
    out = torch.empty(K, dtype=dtype, device=e1.device)
    _hip.call('beer_kl_div', _hip.dtype_code(dtype), K, Q, _hip.ptr(es), _hip.ptr(e1),
              _hip.ptr(e2), _hip.ptr(l1), _hip.ptr(l2), _hip.ptr(out))
    out = out.to(home)
    return out[0] if single else out
