from .basemodel import *
from .modelset import *
from .parameters import *
from .weights import *
from .gaussians import *
from .mixtures import *
from .sequence import *
from .vae import *
