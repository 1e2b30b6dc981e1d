"""Shared helpers for the test-suite (golden loading, oracle import)."""

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import beer_oracle as orc  # noqa: E402  (tests may use the oracle)

STD_NAMES = {
    'NormalWishart': ('mean', 'scale', 'scale_matrix', 'dof'),
    'NormalGamma': ('mean', 'scale', 'shape', 'rates'),
    'IsotropicNormalGamma': ('mean', 'scale', 'shape', 'rate'),
    'Dirichlet': ('concentrations',),
    'Gamma': ('shape', 'rate'),
}
COV_OF = {'NormalWishart': 'full', 'NormalGamma': 'diagonal',
          'IsotropicNormalGamma': 'isotropic'}


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as f:
        return {k: f[k] for k in f.files}


def dist_cls(g, prefix):
    return str(g[prefix + '.cls'])


def std_params(g, prefix):
    'Tuple of standard-parameter arrays of the distribution at `prefix`.'
    return tuple(g[f'{prefix}.{n}'].copy() for n in STD_NAMES[dist_cls(g, prefix)])


def n_params(g, prefix):
    i = 0
    while f'{prefix}.p{i}.prior.cls' in g:
        i += 1
    return i


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    same = (a == b) | (np.isnan(a) & np.isnan(b))     # equal infinities match
    if np.any(~np.isfinite(a) & ~same) or np.any(~np.isfinite(b) & ~same):
        return np.inf
    fin = np.isfinite(b)
    if not fin.any():
        return 0.
    denom = np.maximum(np.abs(b[fin]).max(), 1e-300)
    return np.abs(np.where(same, 0., a - np.where(same, a, b))).max() / denom


def assert_close(a, b, tol, what=''):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, f'{what}: shape {a.shape} != {b.shape}'
    err = rel_err(a, b)
    assert err <= tol, f'{what}: rel err {err:.3e} > {tol:.1e}'


def stat_blocks(Q, D):
    '''Column blocks of accumulated Normal statistics [.., Q] in the reference's
    layout [sum g x | -1/2 second moments | -1/2 N, +1/2 N]
    (beer/dists/normalwishart.py:30-38, normalgamma.py:20-27, isonormalgamma.py:21-30).'''
    return (('first moments', slice(0, D)), ('second moments', slice(D, Q - 2)),
            ('counts', slice(Q - 2, Q)))


def assert_stats_close(got, truth, D, tol, what='', ref32=None, slack=1.):
    '''Accumulated statistics [K, Q] held PER BLOCK: the counts, the first and the
    second moments each against their own largest entry.  A max-norm over the whole
    array would let the counts and first moments be off by (largest second moment /
    their size) x tol.  With `ref32` every block gets assert_within_f32_band's band.'''
    got, truth = np.asarray(got, dtype=np.float64), np.asarray(truth, dtype=np.float64)
    assert got.shape == truth.shape, f'{what}: shape {got.shape} != {truth.shape}'
    errs = {}
    for name, sl in stat_blocks(truth.shape[-1], D):
        if ref32 is None:
            assert_close(got[..., sl], truth[..., sl], tol, f'{what} [{name}]')
            errs[name] = rel_err(got[..., sl], truth[..., sl])
        else:
            errs[name], _ = assert_within_f32_band(got[..., sl], truth[..., sl],
                                                   np.asarray(ref32)[..., sl],
                                                   f'{what} [{name}]', tol=tol, slack=slack)
    return errs


def assert_within_f32_band(got, truth, ref32, what='', tol=1e-5, slack=1.):
    '''float32 results against the fp64 truth of the same float32 inputs: within
    north_star's 1e-5, or -- where float32 arithmetic itself cannot do that --
    within the error of the reference's own float32 op sequence (`ref32`: the
    oracle run in float32, or the reference's float32 golden).  `slack` > 1 only
    where `ref32`'s error is ONE realisation of rounding noise amplified by an
    ill-conditioned step (a posterior scale matrix from less than a frame per
    component), i.e. an estimate of float32's error there, not a bound on it.'''
    band = max(tol, slack * rel_err(ref32, truth))
    err = rel_err(got, truth)
    assert err <= band, f'{what}: rel err {err:.3e} > band {band:.3e} (reference fp32: ' \
                        f'{rel_err(ref32, truth):.3e})'
    return err, band
