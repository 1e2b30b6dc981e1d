'''beer_amd -- MI355X-native variational-Bayes hot path of beer, behind
beer's own model API (`import beer_amd as beer`).'''

from .models import *
from .inference import *
from . import dists
from . import features
from . import graph
from . import nnet
from . import utils
from . import vbi
from .stats import FrameStats
from ._hip import get_f32_mode, set_f32_mode

__version__ = '0.1.0'
