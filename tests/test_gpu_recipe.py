"""The reference's OWN model shapes on the matrix-core path, held to the numpy oracle.

recipes/aud/conf/hmm.yml:3-5, 31-33: a JointModelSet of a 5-state non-speech unit with 10
diagonal Gaussians per state and 3-state speech units with 4 Gaussians per state; features of
recipes/*/conf/mfcc.yml: 39 dimensions (42 with the energy, as bench config 5 extracts them).
>= 16 384 float32 frames (the bf16x3 matrix kernels), free phone loop and alignment graphs,
against orc.hmm_elbo_step per utterance (beer/models/hmm.py:73-100, mixtureset.py:85-112,
modelset.py:71-85): ELBO at 1e-5, the accumulated statistics per block at 1e-5 (or the error of
the oracle's own float32 run where float32 cannot do better).
"""

import os
import sys

import numpy as np
import pytest
import torch

from helpers import assert_close, assert_stats_close, assert_within_f32_band, orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import beer_amd as beer                                             # noqa: E402
from beer_amd import _hip                                            # noqa: E402
from benchlib import recipe                                         # noqa: E402
from gpu_helpers import DEV, npy, tt                                 # noqa: E402
from oracle import graph_oracle as go                               # noqa: E402


def _groups(ploop, dtype=np.float64):
    out = []
    for ms in ploop.modelset.original_modelset.modelsets:
        p, w = ms.modelset.means_precisions, ms.categoricalset.weights
        cast = lambda d: [npy(getattr(d.params, n)).astype(dtype) for n in d._std_params_def]   # noqa: E731
        out.append(dict(cov_type=ms.modelset.cov_type, S=len(ms), G=ms.n_comp_per_mixture,
                        post=cast(p.posterior), prior=cast(p.prior),
                        w_post=cast(w.posterior)[0], w_prior=cast(w.prior)[0]))
    return out


def _corpus(ploop, units, D, n_utts, rng):
    '''Utterances of ~300 frames that follow a unit sequence sil p p ... sil, every frame drawn
    around the mean of a component of the state it is in: posteriors neither flat nor one-hot.'''
    sets = ploop.modelset.original_modelset.modelsets
    mus, first = [], 0
    for ms in sets:
        mu = npy(ms.modelset.means_precisions.posterior.params.mean).astype(np.float64)
        mus.append((first, ms.n_comp_per_mixture, mu))
        first += len(ms)
    names = list(units)
    speech = [n for n in names if not str(n).startswith('sil')]
    utts, seqs = [], []
    for _ in range(n_utts):
        seq = ['sil'] + [speech[i] for i in rng.randint(0, len(speech), rng.randint(6, 12))] + ['sil']
        frames = []
        for name in seq:
            pdfs = [units[name].state_from_id(s).pdf_id for s in units[name].states()]
            for pdf in [p for p in pdfs if p is not None]:
                for f0, G, mu in mus:
                    if f0 <= pdf < f0 + len(mu) // G:
                        n = rng.randint(6, 14)
                        comp = (pdf - f0) * G + rng.randint(0, G, n)
                        frames.append(mu[comp] + rng.randn(n, D) * 1.2)
        utts.append(np.concatenate(frames).astype(np.float32))
        seqs.append(seq)
    return utts, seqs


def _spy():
    calls, orig = [], _hip.call

    def call(name, *a):
        calls.append(name)
        return orig(name, *a)
    return calls, orig, call


@pytest.mark.parametrize('D', [39, 42])
@pytest.mark.parametrize('mode', ['free', 'ali'])
def test_recipe_model_vs_oracle(D, mode):
    rng = np.random.RandomState(D + (mode == 'ali'))
    ploop, units = recipe.phone_loop(40, torch.zeros(D), torch.ones(D), noise_std=1., seed=D)
    ploop = ploop.to(DEV)
    sets = ploop.modelset.original_modelset.modelsets
    assert [(len(m), m.n_comp_per_mixture) for m in sets] == [(5, 10), (120, 4)]
    utts, seqs = _corpus(ploop, units, D, 64, rng)
    lens = [len(u) for u in utts]
    assert sum(lens) >= 16384
    N = 1_000_000
    X = torch.cat([tt(u) for u in utts])
    assert _hip.f32_fast_ok(X)
    graphs = None
    if mode == 'ali':
        graphs = list(beer.graph.compile_alignments(seqs, units))
    calls, orig, spy = _spy()
    _hip.call = spy
    try:
        elbo = beer.accumulate_elbo(ploop, (X, lens), datasize=N, inference_graphs=graphs)
    finally:
        _hip.call = orig
    # the matrix-core kernels are what ran for both groups of Gaussians
    assert 'beer_mixtureset_accumulate_fused' in calls, sorted(set(calls))
    assert 'beer_normal_accumulate' not in calls, sorted(set(calls))

    def oracle(dtype):
        groups = _groups(ploop, dtype)
        cast = lambda a: np.asarray(a).astype(dtype)                                 # noqa: E731
        cat = ploop.categorical.weights
        extra_kl = orc.dir_kl(cast(npy(cat.posterior.params.concentrations)),
                              cast(npy(cat.prior.params.concentrations))).sum()
        value, acc, counts = 0., [[0., 0.] for _ in groups], 0.
        gr = ploop.graph
        loop = dict(init=cast(npy(gr.init_log_probs)), final=cast(npy(gr.final_log_probs)),
                    trans=cast(npy(gr.trans_log_probs)), order=np.asarray(gr.pdf_id_mapping))
        starts, ends = list(ploop.start_pdf.values()), list(ploop.end_pdf.values())
        for u, x in enumerate(utts):
            if mode == 'ali':
                init, final, trans, order = go.compile_graph(
                    go.alignment_graph(seqs[u], units, beer.graph.Graph))
                with np.errstate(divide='ignore'):
                    graph = dict(init=cast(np.log(init)), final=cast(np.log(final)),
                                 trans=cast(np.log(trans)), order=np.asarray(order))
            else:
                graph = loop
            want_xi = mode == 'free' and dtype == np.float64
            r = orc.hmm_elbo_step(x.astype(dtype), groups, graph, datasize=N,
                                  trans_posteriors=want_xi, extra_kl=extra_kl)
            value += r['value']
            for a, (an, aw) in zip(acc, r['acc']):
                a[0], a[1] = a[0] + an, a[1] + aw
            if want_xi:
                counts = counts + orc.cat_suffstats(orc.phone_counts(
                    r['trans_resps'], r['resps'], starts, ends).reshape(1, -1)).sum(0)
        return value, acc, counts
    value, acc, counts = oracle(np.float64)
    _, acc32, _ = oracle(np.float32)
    assert_close(float(elbo), value, 1e-5, 'elbo')
    for ms, (an, aw), (an32, aw32) in zip(sets, acc, acc32):
        tag = f'S={len(ms)} G={ms.n_comp_per_mixture}'
        assert_stats_close(npy(elbo._acc_stats[ms.modelset.means_precisions]), an, D, 1e-5,
                           f'statistics of the Gaussians ({tag})', ref32=an32)
        assert_within_f32_band(npy(elbo._acc_stats[ms.categoricalset.weights]).astype(np.float64),
                               aw, aw32, f'statistics of the weights ({tag})')
    if mode == 'free':
        assert_close(npy(elbo._acc_stats[ploop.categorical.weights]), counts, 1e-5, 'phone counts')
    # every frame's posteriors sum to one: the counts of the groups add up to the frames
    total = sum(float(npy(elbo._acc_stats[ms.categoricalset.weights]).astype(np.float64)[:, -1].sum())
                for ms in sets)
    assert abs(total - sum(lens)) <= 1e-6 * sum(lens)
