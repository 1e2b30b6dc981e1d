'Command-line front end (`beer <cmd> <subcmd>`), data set format, pickle compatibility.'
from .dataset import Dataset, Utterance
from . import compat
