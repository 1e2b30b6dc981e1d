'''beer_amd -- MI355X-native variational-Bayes hot path of beer, behind
beer's own model API (`import beer_amd as beer`).'''

import os as _os

import torch as _torch

# The host side of this package only does small bookkeeping with CPU tensors
# (offset tables of a batch, a few hundred floats).  torch parallelises even
# those over every core: on a 256-thread host the OpenMP fork/join of a 100 KB
# copy was measured to stall an iteration by 30-100 ms.  Cap the intra-op pool
# (BEER_HOST_THREADS overrides; bench.py's CPU baseline sets its own count).
_torch.set_num_threads(int(_os.environ.get('BEER_HOST_THREADS',
                                           min(_torch.get_num_threads(), 4))))

from .models import *
from .inference import *
from . import dists
from . import features
from . import graph
from . import nnet
from . import utils
from . import vbi
from .stats import FrameImages, FrameStats, reference_layout
from ._hip import exact_f32, get_f32_mode, set_f32_mode

__version__ = '0.1.0'
