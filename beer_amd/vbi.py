"""`beer.vbi` is dead legacy code in the reference snapshot (SURVEY.md section
0 fact 3: its optimizers call attributes that no longer exist).  The north
star names it, so this module offers the names that still make sense as thin
aliases of the live API (beer/inference)."""

from .inference.objectives import EvidenceLowerBoundInstance, evidence_lower_bound
from .inference.optimizers import VBConjugateOptimizer, VBOptimizer

BayesianModelOptimizer = VBConjugateOptimizer
__all__ = ['evidence_lower_bound', 'EvidenceLowerBoundInstance', 'VBConjugateOptimizer',
           'VBOptimizer', 'BayesianModelOptimizer']
