from .objectives import *
from .optimizers import *
from .batch import *
from .captured import *
