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

import pickle

__all__ = ['VBConjugateOptimizer', 'VBOptimizer']

_STATE = ('lrate', 'update_count')


class VBConjugateOptimizer:
    'Round-robin coordinate ascent over mean-field groups of conjugate parameters.'

    def __init__(self, groups, lrate=1.):
        # a model may hand its groups over as generators: materialise them once
        self.groups = [list(members) for members in groups]
        self.lrate, self.update_count = lrate, 0

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
        for param in self._every_parameter():
            param.zero_stats()

    def step(self):
        'Natural-gradient update of the group whose turn it is.'
        for param in self._group_in_turn():
            param.natural_grad_update(self.lrate)
        self.update_count += 1


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
