"""Drivers of the M-step.

`VBConjugateOptimizer` walks the mean-field groups of a model round-robin -- one
group per `step()` gets its natural-gradient update (the kernels behind
`ConjugateBayesianParameter.natural_grad_update`), the others wait for their turn
(SURVEY.md appendix B, quirk Q7).  `VBOptimizer` pairs it with a torch optimizer
for the parameters that are not conjugate (the networks of a VAE).

Interface of beer/inference/optimizers.py:5-67 (constructor arguments,
`init_step / step / state_dict / load_state_dict / load_state`, the attributes
`groups`, `lrate`, `update_count`, `cjg_optim`, `std_optim` that pickled
optimizer states carry).
"""

import os
import pickle
import warnings

import torch

from .. import _hip

__all__ = ['VBConjugateOptimizer', 'VBOptimizer']

_STATE = ('lrate', 'update_count')


class VBConjugateOptimizer:
    'Round-robin coordinate ascent over mean-field groups of conjugate parameters.'

    _warned = False                               # (one warning per process about a refused capture)

    def __init__(self, groups, lrate=1., graph=None):
        # a model may hand its groups over as generators: materialise them once
        self.groups = [list(members) for members in groups]
        self.lrate, self.update_count = lrate, 0
        # The update of a group is captured once as a HIP graph and replayed -- one launch
        # instead of ~20 per parameter -- whenever the group can be captured (device tensors,
        # device-only callbacks: `_capturable`); other groups, and every group with
        # graph=False or BEER_MSTEP_GRAPH=0, take the eager update.
        # IN-PLACE semantics of the replayed update: the reference replaces the posterior's
        # tensors at every update (parameters.py:134-141); a replay rewrites the tensors of the
        # capture instead.  Code that keeps references to a posterior's tensors across updates
        # (a convergence check, a snapshot) sees them change: `.clone()` what has to stay, or
        # construct with graph=False for the reference's replace-on-update behaviour.  The first
        # update of every group synchronises the device once (the capture).
        self.graph = (os.environ.get('BEER_MSTEP_GRAPH', '1') != '0') if graph is None \
            else bool(graph)
        self._captured = {}

    def __getstate__(self):
        state = self.__dict__.copy()
        state['_captured'] = {}                   # graphs belong to the process that made them
        return state

    # -- persistent state: the learning rate and whose turn it is
    def state_dict(self):
        return {key: getattr(self, key) for key in _STATE}

    def load_state_dict(self, state_dict):
        for key in _STATE:
            setattr(self, key, state_dict[key])

    def _every_parameter(self):
        return (param for members in self.groups for param in members)

    def _group_in_turn(self):
        return self.groups[self.update_count % len(self.groups)] if self.groups else ()

    def init_step(self):
        'Forget the statistics of the previous iteration.'
        params = list(self._every_parameter())
        stats = [getattr(p, 'stats', None) for p in params]
        if len(params) > 1 and all(isinstance(t, torch.Tensor) and t.is_cuda for t in stats) and \
                all(type(p).zero_stats is type(params[0]).zero_stats for p in params):
            # (one launch for all of them: at the notebook's sizes an iteration is ~30 launches of
            #  ~4.5 us each, whatever they do)
            torch._foreach_zero_(stats)
            return
        for param in params:
            param.zero_stats()

    def step(self):
        'Natural-gradient update of the group whose turn it is.'
        members = self._group_in_turn()
        if not (self.graph and members and self._replay(members)):
            for param in members:
                param.natural_grad_update(self.lrate)
        self.update_count += 1

    # -- the M-step of a group as a captured HIP graph ---------------------------
    # The update of a parameter is ~20 small launches (natural-gradient step, eta -> standard
    # parameters with a D x D factorisation per Gaussian, E[T], log-normaliser); their GPU
    # time hides behind the E-step, their host time does not.  Captured once per group,
    # the kernels read the statistics and the current eta from buffers that stay where they
    # are and rewrite the posterior's tensors in place; after a replay the posterior's memo
    # is put back to what the capture produced (same tensors, new contents) and everything
    # computed lazily since (KL terms, log-weights ...) is dropped.
    @staticmethod
    def _signature(members):
        'Identity and version of every tensor of the members\' posteriors and priors.'
        return tuple((id(t), t._version) for p in members
                     for t in tuple(p.posterior._tensors()) + tuple(p.prior._tensors()))

    @staticmethod
    def _capturable(members):
        for p in members:
            if getattr(p, '_callbacks', None) and not p.callbacks_device_only():
                return False                      # callbacks that run host code / host copies
            try:
                tensors = tuple(p.posterior._tensors()) + tuple(p.prior._tensors()) + (p.stats,)
            except AttributeError:
                return False                      # (not a ConjugateBayesianParameter: eager)
            if not all(isinstance(t, torch.Tensor) and t.is_cuda for t in tensors):
                return False
        return True

    def _replay(self, members):
        key = self.update_count % len(self.groups)
        entry = self._captured.get(key)
        if entry is False:
            return False
        if entry is not None and (entry[4] != self.lrate or entry[5] != self._signature(members)):
            # a new learning rate, or the posterior was touched from outside since the
            # capture (a state-dict load, another optimizer, a re-initialisation: its tensors
            # were replaced or written in place -- a replay itself leaves torch's version
            # counters alone): the graph's private eta is stale, capture again
            entry = None
        if entry is None:
            if not self._capturable(members):
                self._captured[key] = False
                return False
            statics = [p.stats.detach().clone() for p in members]
            etas = [p.posterior.natural_parameters().detach().clone() for p in members]
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            try:
                # (thread-local capture: other threads of the process -- a collective's
                # watchdog -- may keep calling into the runtime)
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    for p, st, eta in zip(members, statics, etas):
                        p.stats = st
                        eta.copy_(p.natural_grad_update(self.lrate, eta_q=eta))
            except RuntimeError as err:
                # Only a REFUSED capture is answered by the eager update: torch's own
                # RuntimeError (an operation that cannot be recorded, a capture next to a
                # collective's watchdog), BEER_EINVAL (`HipInvalid`) or one of HIP's
                # stream-capture status codes (900 .. 908) out of the library.  Anything else
                # from the library is a launch error and propagates (`_hip.HipError`).
                refused = not isinstance(err, _hip.HipError) or isinstance(err, _hip.HipInvalid) \
                    or 900 <= -(err.rc or 0) <= 908
                # Nothing ran on the device -- a capture records, it does not execute --, so
                # the eager update starts from the same posterior; the host-side memos the
                # recording wrote have to go.
                for p, st in zip(members, statics):
                    p.stats = st
                    p.posterior.__dict__.pop('_memo', None)
                    p.__dict__.pop('_kl_memo', None)
                if not refused or os.environ.get('BEER_MSTEP_GRAPH_STRICT') == '1':
                    raise
                self._captured[key] = False
                if not VBConjugateOptimizer._warned:
                    VBConjugateOptimizer._warned = True
                    warnings.warn(f'VBConjugateOptimizer: the M-step of group {key} could not be '
                                  f'captured as a HIP graph ({type(err).__name__}: {err}); this '
                                  'group takes the eager update from here on', RuntimeWarning)
                return False
            memos = [dict(p.posterior.__dict__.get('_memo', {})) for p in members]
            entry = self._captured[key] = (graph, statics, etas, memos, self.lrate,
                                           self._signature(members))
        else:
            graph, statics, etas, memos = entry[:4]
            for p, st in zip(members, statics):
                if p.stats is not st:
                    st.copy_(p.stats)
                    p.stats = st
        graph, statics, etas, memos = entry[:4]
        graph.replay()
        for p, memo in zip(members, memos):
            p.posterior.__dict__['_memo'] = dict(memo)
            p.__dict__.pop('_kl_memo', None)
        return True


class VBOptimizer:
    '''A conjugate optimizer and / or a torch optimizer behind one
    `init_step / step`: gradient step first, then the conjugate update.'''

    def __init__(self, cjg_optim=None, std_optim=None):
        self.cjg_optim, self.std_optim = cjg_optim, std_optim

    def _parts(self):
        'The optimizers that are present, with the key their state is filed under.'
        return [(key, getattr(self, key)) for key in ('cjg_optim', 'std_optim')
                if getattr(self, key) is not None]

    def state_dict(self):
        return {key: optim.state_dict() for key, optim in self._parts()}

    def load_state_dict(self, state_dict):
        for key, optim in self._parts():
            optim.load_state_dict(state_dict[key])

    def load_state(self, path):
        '''Learning rate and update count from a pickled `(lrate, update_count)`
        pair (what the reference's method of this name reads).'''
        with open(path, 'rb') as f:
            self.lrate, self.update_count = pickle.load(f)

    def init_step(self):
        if self.cjg_optim is not None:
            self.cjg_optim.init_step()
        if self.std_optim is not None:
            self.std_optim.zero_grad()

    def step(self):
        for optim in (self.std_optim, self.cjg_optim):
            if optim is not None:
                optim.step()
