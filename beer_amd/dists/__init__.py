'''Exponential-family priors / posteriors and their conjugate likelihoods.'''

from .expfam import *
from .families import *
from .normaldiag import *
