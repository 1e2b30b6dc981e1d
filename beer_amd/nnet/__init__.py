'''Small torch.nn building blocks used by the VAE examples of the reference
(beer/nnet/residual.py).  Plain torch: the networks are not the hot path.'''

from .linear import *
from .residual import *
