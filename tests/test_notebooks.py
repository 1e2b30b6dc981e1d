"""The reference's example notebooks with `import beer_amd as beer`: the same call
sequences (tests/golden/notebook_cells.py: Mixture Model.ipynb, HMM.ipynb,
HMM_align.ipynb) against what the reference produced for them
(tests/golden/g16_notebooks.npz, written by make_golden.py).  Models stay where
the notebooks create them -- on the host, in float64; the frames are host
tensors: the drop-in layer moves bytes, the kernels compute."""

import os
import sys

import numpy as np
import pytest

from helpers import GOLDEN, assert_close, load_golden

pytestmark = pytest.mark.gpu

sys.path.insert(0, GOLDEN)
import beer_amd as beer                      # noqa: E402
import notebook_cells as nb                  # noqa: E402

TOL = 1e-8


@pytest.mark.parametrize('variant', ['dirichlet', 'sb', 'sb_hyper'])
def test_mixture_model_notebook(variant):
    g = load_golden('g16_notebooks')
    got = nb.mixture_model(beer, g['mixture.data'], variant, epochs=8)
    for key, val in got.items():
        ref = g[f'mixture.{variant}.{key}']
        if key == 'ordering':
            np.testing.assert_array_equal(val, ref)
        elif key == 'repr_nonempty':
            assert bool(val)
        else:
            assert_close(val, ref, TOL, f'{variant}.{key}')


def test_hmm_notebook():
    g = load_golden('g16_notebooks')
    got = nb.hmm(beer, g['hmm.data'], epochs=8)
    for key, val in got.items():
        ref = g[f'hmm.{key}']
        if key == 'best_path':
            np.testing.assert_array_equal(val, ref)
        else:
            assert_close(val, ref, TOL, key)


def test_hmm_align_notebook():
    g = load_golden('g16_notebooks')
    got = nb.hmm_align(beer, g['align.data'], epochs=6)
    for key, val in got.items():
        assert_close(val, g[f'align.{key}'], 1e-7, key)
