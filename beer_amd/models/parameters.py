"""Bayesian parameters (prior + posterior + accumulated statistics).

API mirror of beer/models/parameters.py:11-141; the natural-gradient update
runs `beer_natural_grad_step` + the family's `*_from_natural` kernel.
"""

import uuid

import torch

from .. import _hip
from ..dists import kl_div

__all__ = ['ParameterRef', 'BayesianParameter', 'ConjugateBayesianParameter']


class ParameterRef:
    '''The identity of a parameter without its tensors: hashes and compares like
    the parameter with the same uuid.  What a pickled ELBO object keys its
    accumulated statistics with (`beer hmm accumulate` -> `beer hmm update`): the
    file then holds no device tensors and no model objects.'''
    __slots__ = ('uuid',)

    def __init__(self, uuid):
        self.uuid = uuid

    def __hash__(self):
        return hash(self.uuid)

    def __eq__(self, other):
        return getattr(other, 'uuid', None) == self.uuid

    def __getstate__(self):
        return {'uuid': self.uuid}

    def __setstate__(self, state):
        self.uuid = state['uuid']

    def __repr__(self):
        return f'ParameterRef({self.uuid})'


class BayesianParameter(torch.nn.Module):
    'A parameter with a prior and a (variational) posterior distribution.'

    def __init__(self, prior, posterior=None):
        super().__init__()
        self.prior = prior
        self.posterior = posterior
        self.uuid = uuid.uuid4()
        self._callbacks = set()

    def __len__(self):
        return len(self.prior)

    def __getitem__(self, key):
        return self.__class__(prior=self.prior[key], posterior=self.posterior[key])

    def __repr__(self):
        post = self.posterior.__class__.__qualname__ if self.posterior is not None \
            else '<unspecified>'
        return (f'{self.__class__.__qualname__}(prior={self.prior.__class__.__qualname__}, '
                f'posterior={post})')

    # Parameters are dictionary keys of the accumulated statistics; the uuid
    # survives pickling (elbo.sync, beer/inference/objectives.py:109-116).
    def __hash__(self):
        return hash(self.uuid)

    def __eq__(self, other):
        return isinstance(other, (BayesianParameter, ParameterRef)) and self.uuid == other.uuid

    def dispatch(self, before_update=False):
        'Run the registered callbacks of the given phase.'
        for callback, notify_before_update in list(self._callbacks):
            if notify_before_update == before_update:
                callback()

    def register_callback(self, callback, notify_before_update=False):
        '''`callback()` runs before / after every update of this parameter
        (beer/models/parameters.py:49-57).  A callback whose work is launches on
        device tensors only may carry the attribute `device_only = True`
        (a function returning a bool: asked each time): the update of its
        group can then be captured and replayed as a HIP graph (optimizers.py).'''
        self._callbacks.add((callback, notify_before_update))

    def callbacks_device_only(self):
        'True when every registered callback declares itself capturable.'
        for callback, _ in self._callbacks:
            flag = getattr(callback, 'device_only', None)
            if flag is None and hasattr(callback, '__func__'):
                flag = getattr(callback.__func__, 'device_only', None)
            if flag is None:
                return False
            if callable(flag):
                owner = getattr(callback, '__self__', None)
                flag = flag(owner) if owner is not None else flag()
            if not flag:
                return False
        return True

    def value(self):
        return self.posterior.expected_value()

    def kl_div_posterior_prior(self):
        '''KL(q || p), one value per pdf.  Iteration-invariant, so it is
        computed once per parameter version rather than once per utterance
        (46 % of the reference's HMM time).'''
        sig = tuple((t, t._version) for t in
                    self.posterior._tensors() + self.prior._tensors())
        memo = self.__dict__.get('_kl_memo')
        if memo is not None and len(memo[0]) == len(sig) and \
                all(a[0] is b[0] and a[1] == b[1] for a, b in zip(memo[0], sig)):
            return memo[1]
        value = kl_div(self.posterior, self.prior)
        self.__dict__['_kl_memo'] = (sig, value)
        return value

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop('_kl_memo', None)
        return state


class ConjugateBayesianParameter(BayesianParameter):
    'Parameter whose likelihood is conjugate to its prior.'

    def __init__(self, prior, posterior, init_stats=None, likelihood_fn=None):
        super().__init__(prior, posterior)
        if init_stats is None:
            ref = prior._tensors()[0]
            init_stats = torch.zeros(prior.natural_shape(), dtype=ref.dtype, device=ref.device)
        self.register_buffer('stats', init_stats.clone().detach())
        if likelihood_fn is None:
            likelihood_fn = prior.conjugate()
        self.likelihood_fn = likelihood_fn

    def __len__(self):
        return 1 if self.stats.dim() <= 1 else self.stats.shape[0]

    def __getitem__(self, key):
        return self.__class__(prior=self.prior[key], posterior=self.posterior[key],
                              init_stats=self.stats[key], likelihood_fn=self.likelihood_fn)

    def zero_stats(self):
        self.stats.zero_()

    def store_stats(self, acc_stats):
        self.stats = acc_stats.clone().detach() if acc_stats.requires_grad else acc_stats

    def natural_form(self):
        'E_q[T(theta)] -- what the E-step kernels consume.'
        return self.posterior.expected_sufficient_statistics()

    def natural_grad_update(self, lrate, eta_q=None):
        '''eta <- eta_q + lrate (eta_p + stats - eta_q); then eta -> std params.  Returns the
        new eta.  `eta_q`: the posterior's natural parameters held by the caller (a
        captured M-step keeps them in a buffer of its own, see optimizers.py).'''
        self.dispatch(before_update=True)
        eta_p = self.prior.natural_parameters()
        if eta_q is None:
            eta_q = self.posterior.natural_parameters()
        home, dtype = eta_q.device, eta_q.dtype
        dp, dq = _hip.on_device(eta_p, dtype), _hip.on_device(eta_q)
        ds = _hip.on_device(self.stats, dtype)
        if ds.shape != dq.shape:
            raise ValueError(f'statistics {tuple(ds.shape)} do not match the natural '
                             f'parameters {tuple(dq.shape)}')
        new = torch.empty_like(dq)
        _hip.call('beer_natural_grad_step', _hip.dtype_code(dtype), dq.numel(),
                  _hip.ptr(dp), _hip.ptr(dq), _hip.ptr(ds), float(lrate), _hip.ptr(new))
        self.posterior.update_from_natural_parameters(new.to(home))
        self.dispatch(before_update=False)
        return new
