"""A whole VB iteration as ONE submission.

The reference's training loop on a resident minibatch (examples/Mixture Model.ipynb cell 9;
beer/inference/objectives.py:119-190, optimizers.py:22-31) is

    optim.init_step()
    elbo = beer.evidence_lower_bound(model, X, datasize=N)
    elbo.backward()
    optim.step()

-- some forty small launches for a mixture of a few Gaussians, each a few microseconds
of GPU time behind tens of microseconds of host time: at BASELINE config 1 (K = 8, D = 2,
1000 frames) the iteration is bound by the host, not by the device.  `CapturedIteration`
records that sequence once as a HIP graph (`torch.cuda.CUDAGraph`: torch as the stream /
graph container) and replays it: one launch per iteration.

What makes the recorded sequence a fixed point.  The eager update REPLACES the posterior's
tensors (as the reference does, parameters.py:134-141); a recorded E-step would go on
reading the tensors it was recorded with.  The capture therefore ends with device copies
of the new posterior back into the tensors the E-step read, and the model keeps those
tensors: every replay reads the previous replay's result.  Everything derived from the
posterior (E[T], log-normalisers, log-weights, KL) is recomputed INSIDE the recording --
the memos are dropped before it and after every replay.  One graph per mean-field group in
turn (optimizers.py:29-31: round robin).
"""

import warnings

import torch

from .. import _hip
from .objectives import evidence_lower_bound
from .optimizers import VBConjugateOptimizer

__all__ = ['CapturedIteration']


def _drop_memos(params):
    for p in params:
        p.posterior.__dict__['_memo'] = {}
        p.__dict__.pop('_kl_memo', None)


class CapturedIteration:
    '''`it = CapturedIteration(model, optim, X, datasize=N, **kwargs)`; every `it()` is one
    VB iteration on the resident minibatch `X` and returns the ELBO value (0-dim float64
    device tensor, overwritten by the next call).  The first call of a mean-field group
    runs eagerly (it settles the allocations and the constants), the second records,
    later ones replay.  `mode` says what the last call did: 'eager', 'captured' or
    'replayed'; a sequence that cannot be recorded (host callbacks, a host -> device
    copy inside the E-step) stays eager, with one warning.'''

    def __init__(self, model, optim, data, datasize=-1, **kwargs):
        if not isinstance(optim, VBConjugateOptimizer):
            raise TypeError('CapturedIteration drives a VBConjugateOptimizer')
        self.model, self.optim = model, optim
        # `data`: one resident minibatch [T, D] (-> evidence_lower_bound), or a shard of
        # utterances as `(X_packed, lengths)` (-> accumulate_elbo: the batched E-step of
        # `beer hmm accumulate` + `update`, accumulate.py:39-63, update.py:41-62).  The shard
        # form keeps what depends on the lengths only (`ShardStatics`) and, for diagonal
        # emissions, the frame fragment images (`FrameImages`) for as long as this object lives.
        self.lengths = None
        if isinstance(data, tuple) and len(data) == 2 and isinstance(data[0], torch.Tensor):
            from ..stats import FrameImages
            from .batch import ShardStatics
            self.data, self.lengths = _hip.on_device(data[0]), [int(n) for n in data[1]]
            kwargs.setdefault('statics', ShardStatics())
            if 'frame_images' not in kwargs and self.data.dtype == torch.float32:
                kwargs['frame_images'] = FrameImages(self.data)
        else:
            self.data = _hip.on_device(data)
        self.datasize, self.kwargs = datasize, kwargs
        self._entries = {}                 # turn -> None (warmed up) | entry
        self._all_eager = False            # a group that cannot be recorded: every turn eager
        self.mode = None

    # -- the reference's loop body ------------------------------------------------------
    def _iteration(self):
        self.optim.init_step()
        if self.lengths is None:
            elbo = evidence_lower_bound(self.model, self.data, datasize=self.datasize,
                                        **self.kwargs)
        else:
            from .batch import accumulate_elbo
            elbo = accumulate_elbo(self.model, (self.data, self.lengths), datasize=self.datasize,
                                   **self.kwargs)
        elbo.backward()
        self.optim.step()
        return elbo

    def _eager(self):
        self.mode = 'eager'
        value = self._iteration().value
        return value if isinstance(value, torch.Tensor) else torch.as_tensor(value)

    def _device_images(self):
        """The device images a recorded E-step reads besides the posteriors: the CSR copy of
        every compiled graph the model (or the call's `inference_graphs`) holds.  A rebuilt image
        is a new object; the recording would go on reading the old one."""
        from ..graph import CompiledGraph

        def image(g):
            return vars(g).get('_device_memo', (None,) * 4)[3]
        images = [image(g) for m in self.model.modules() for g in vars(m).values()
                  if isinstance(g, CompiledGraph)]
        extra = self.kwargs.get('inference_graphs')
        if extra is not None and len(extra):
            owners = {id(o): o for o in (getattr(extra[0], '_set', None), getattr(extra[-1], '_set', None))
                      if o is not None}
            if owners:                     # graphs of a GraphSet: one blob per set and dtype
                images += [len(extra)] + [blob for o in owners.values() for blob, _ in o._images.values()]
            else:
                images += [image(g) for g in extra if isinstance(g, CompiledGraph)]
        return tuple(images)

    def _signature(self):
        """What every recording of this object depends on: the learning rate, the data, the
        posterior tensors of EVERY mean-field group (the E-step of one group's turn reads all of
        them) and the graphs' device images.  The tuple holds the objects themselves -- compared by
        identity, and kept alive so that no identity can be reused."""
        every = [p for group in self.optim.groups for p in group]
        return (self.optim.lrate, self.data.data_ptr(), tuple(self.data.shape)) + \
            tuple(t for p in every for t in p.posterior._tensors()) + self._device_images()

    @staticmethod
    def _same(a, b):
        return len(a) == len(b) and a[:3] == b[:3] and \
            all(x is y or (isinstance(x, int) and x == y) for x, y in zip(a[3:], b[3:]))

    def __call__(self):
        optim = self.optim
        if not optim.groups:
            return self._eager()
        turn = optim.update_count % len(optim.groups)
        members = optim.groups[turn]
        entry = self._entries.get(turn, 'new')
        if self._all_eager:
            return self._eager()
        if entry == 'new':
            self._entries[turn] = None
            return self._eager()
        if entry is not None and not self._same(entry['signature'], self._signature()):
            # a posterior of ANY group replaced from outside (or by a group's eager warm-up turn),
            # a rebuilt graph image, a new learning rate: every recording read the old objects
            self._entries = {t: None for t in self._entries}
            entry = None
        if entry is None:
            entry = self._capture(turn, members)
            if entry is None:
                return self._eager()
            self.mode = 'captured'
        else:
            self.mode = 'replayed'
        entry['graph'].replay()
        optim.update_count += 1
        every = [p for group in optim.groups for p in group]
        _drop_memos(every)
        for p, st in zip(members, entry['stats']):
            p.stats = st
        return entry['value']

    def _capture(self, turn, members):
        optim = self.optim
        every = [p for group in optim.groups for p in group]
        # EVERY group has to be recordable: a group that stays eager (host callback, host graph
        # tensors, non-conjugate parameter) replaces its posterior tensors each turn, and the other
        # groups' recordings would go on reading the old -- possibly freed -- ones.
        if not all(VBConjugateOptimizer._capturable(group) for group in optim.groups) or \
                not all(t.is_cuda for p in every for t in p.posterior._tensors()):
            self._all_eager = True
            return None
        homes = [p.posterior.params for p in members]
        home_tensors = [tuple(p.posterior._tensors()) for p in members]
        count, use_graph = optim.update_count, optim.graph
        _drop_memos(every)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        try:
            optim.graph = False            # (the M-step is part of THIS recording)
            with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                elbo = self._iteration()
                value = elbo.value
                # (the group's new posterior back into the tensors the E-step read: ONE launch)
                dsts = [dst for old in home_tensors for dst in old]
                srcs = [src for p in members for src in p.posterior._tensors()]
                torch._foreach_copy_(dsts, srcs)
        except RuntimeError as err:
            refused = not isinstance(err, _hip.HipError) or isinstance(err, _hip.HipInvalid) \
                or 900 <= -(err.rc or 0) <= 908
            if not refused:
                raise
            self._all_eager = True
            warnings.warn(f'CapturedIteration: the iteration could not be recorded as a HIP graph '
                          f'({type(err).__name__}: {err}); it runs eagerly', RuntimeWarning)
            return None
        finally:
            optim.graph, optim.update_count = use_graph, count
            # nothing has run: the model is the one the recording started from
            for p, params in zip(members, homes):
                p.posterior.params = params
            _drop_memos(every)
        entry = {'graph': graph, 'value': value, 'stats': [p.stats for p in members],
                 'signature': self._signature()}
        self._entries[turn] = entry
        return entry
