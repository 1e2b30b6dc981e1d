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


def test_hmm_vae_notebook(monkeypatch):
    '''examples/HMM-VAE.ipynb cells 2-5 and 7-9 (`notebook_cells.hmm_vae`) with
    `import beer_amd as beer`: the HMM trained alone, then a VAE around it --
    `VBOptimizer(VBConjugateOptimizer(lrate=0), Adam)`, `evidence_lower_bound(vae, X, nsamples=5)`
    for 8 epochs, the prior's learning rate switched on after 3 -- from the reference's
    initial network weights and the noise its `posts.sample(5)` drew (g19_hmm_vae_notebook.npz,
    make_golden.py: g19_hmm_vae_notebook).  The reference's ELBO is its [T, 1] - [T] broadcast
    (vae.py:84-86): T times the per-frame sum, which `beer_amd.VAE` returns by default.'''
    import contextlib
    import torch
    from beer_amd.dists import normaldiag
    g = load_golden('g19_hmm_vae_notebook')
    nn_init = {k[len('nn_init.'):]: g[k] for k in g if k.startswith('nn_init.')}
    draws = iter(g['noise'])

    def replay(*shape, **conf):
        t = torch.from_numpy(next(draws).copy())
        return t.to(dtype=conf.get('dtype', t.dtype), device=conf.get('device', 'cpu'))
    monkeypatch.setattr(normaldiag, '_randn', replay)
    got = nb.hmm_vae(beer, g['data'], hmm_epochs=5, epochs=8, update_prior_after_epoch=3,
                     randomness=contextlib.nullcontext(), nn_init=nn_init)
    assert_close(got['hmm.elbos'], g['hmm.elbos'], TOL, 'HMM alone')
    assert_close(got['elbos'], g['elbos'], 1e-7, 'VAE ELBO per epoch')
    for key in got:
        if key.startswith('nn_final.') or key.startswith('prior.'):
            assert_close(got[key], g[key], 1e-6, key)
