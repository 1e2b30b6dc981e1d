"""Command-line front end, data set format and pickle compatibility with the
reference.  Building graphs / loading pickles is host logic (CPU); the
accumulate / update / decode commands need the GPU."""

import io
import logging
import os
import pickle
import sys

import numpy as np
import pytest
import torch

from helpers import GOLDEN, assert_close, load_golden, std_params

import beer_amd as beer
from beer_amd.cli import compat, main as cli_main
from beer_amd.cli.dataset import Dataset

LOG = logging.getLogger('test')

HMM_CONF = """
- group_name: sil
  n_normal_per_state: 3
  prior_strength: 1.
  noise_std: 0.5
  cov_type: diagonal
  shared_cov: no
  topology:
  - {start_id: 0, end_id: 1, trans_prob: 1.0}
  - {start_id: 1, end_id: 1, trans_prob: 0.5}
  - {start_id: 1, end_id: 2, trans_prob: 0.5}
  - {start_id: 2, end_id: 2, trans_prob: 0.5}
  - {start_id: 2, end_id: 1, trans_prob: 0.25}
  - {start_id: 2, end_id: 3, trans_prob: 0.25}
- group_name: speech
  n_normal_per_state: 4
  prior_strength: 1.
  noise_std: 0.5
  cov_type: diagonal
  shared_cov: no
  topology:
  - {start_id: 0, end_id: 1, trans_prob: 1.0}
  - {start_id: 1, end_id: 1, trans_prob: 0.75}
  - {start_id: 1, end_id: 2, trans_prob: 0.25}
  - {start_id: 2, end_id: 2, trans_prob: 0.75}
  - {start_id: 2, end_id: 3, trans_prob: 0.25}
  - {start_id: 3, end_id: 3, trans_prob: 0.75}
  - {start_id: 3, end_id: 4, trans_prob: 0.25}
"""
UNITS = 'sil sil\na speech\nb speech\nc speech\nd speech\n'


def run(argv, stdin=''):
    old_in, old_out = sys.stdin, sys.stdout
    sys.stdin, sys.stdout = io.StringIO(stdin), io.StringIO()
    try:
        cli_main.main(argv)
        return sys.stdout.getvalue()
    finally:
        sys.stdin, sys.stdout = old_in, old_out


def test_dataset_create_and_pickle(tmp_path):
    rng = np.random.RandomState(0)
    feats = {f'u{i}': rng.randn(T, 3).astype(np.float32) for i, T in enumerate((7, 11))}
    npz = tmp_path / 'feats.npz'
    np.savez(npz, **feats)
    run(['dataset', 'create', str(tmp_path), str(npz), str(tmp_path / 'ds.pkl')])
    ds = pickle.load(open(tmp_path / 'ds.pkl', 'rb'))
    allx = np.concatenate(list(feats.values()))
    assert ds.size == 18 and len(ds) == 2
    assert_close(ds.mean.numpy(), allx.mean(0), 1e-6)
    assert_close(ds.var.numpy(), allx.var(0), 1e-5)
    utt = ds['u1']
    assert utt.id == 'u1' and utt.features.dtype == torch.float32 and len(utt.features) == 11
    assert [u.id for u in ds.utterances(random_order=False)] == ['u0', 'u1']


def test_graph_building_commands_match_reference(tmp_path):
    (tmp_path / 'hmm.yml').write_text(HMM_CONF)
    (tmp_path / 'units').write_text(UNITS)
    run(['-s', '1', 'hmm', 'mkphones', '-D', '3', str(tmp_path / 'hmm.yml'),
         str(tmp_path / 'units'), str(tmp_path / 'hmms.mdl')])
    run(['hmm', 'mkphoneloopgraph', '--start-end-group', 'sil', str(tmp_path / 'units'),
         str(tmp_path / 'ploop_graph.pkl')])
    run(['hmm', 'mkdecodegraph', str(tmp_path / 'ploop_graph.pkl'), str(tmp_path / 'hmms.mdl'),
         str(tmp_path / 'decode_graph.pkl')])
    units, emissions = pickle.load(open(tmp_path / 'hmms.mdl', 'rb'))
    assert list(units) == ['sil', 'a', 'b', 'c', 'd'] and len(emissions) == 14
    assert [len(m) for m in emissions.modelsets] == [2, 12]
    assert [m.n_comp_per_mixture for m in emissions.modelsets] == [3, 4]
    graph, start_pdf, end_pdf = pickle.load(open(tmp_path / 'decode_graph.pkl', 'rb'))
    g = load_golden('g12_graph_compile')
    assert list(start_pdf.values()) == g['start_idxs'].tolist()
    assert list(end_pdf.values()) == g['end_idxs'].tolist()
    cg = graph.compile()
    assert cg.pdf_id_mapping == g['ploop.pdf_id_mapping'].tolist()
    assert_close(cg.init_log_probs.exp().numpy(), np.exp(g['ploop.init']), 1e-6)
    # alignment graphs: np.save of an object array [CompiledGraph], as the reference writes
    os.makedirs(tmp_path / 'ali')
    run(['hmm', 'mkaligraph', str(tmp_path / 'hmms.mdl'), str(tmp_path / 'ali')],
        stdin='utt1 sil a c a sil\nutt2\n')
    ali = np.load(tmp_path / 'ali' / 'utt1.npy', allow_pickle=True)[0]
    assert ali.pdf_id_mapping == g['ali.pdf_id_mapping'].tolist()
    assert_close(ali.trans_log_probs.exp().numpy(), np.exp(g['ali.trans']), 1e-6)
    assert run(['hmm', 'phonelist', str(tmp_path / 'hmms.mdl')]).split() == \
        ['a', 'b', 'c', 'd', 'sil']


def test_reference_pickles_load_into_beer_amd_classes():
    ploop = compat.load(open(os.path.join(GOLDEN, 'ref_phoneloop.pkl'), 'rb'))
    assert type(ploop) is beer.PhoneLoop
    assert type(ploop.graph) is beer.graph.CompiledGraph
    groups = ploop.mean_field_factorization()
    assert len(groups) == 1 and len(groups[0]) == 5
    ns = ploop.modelset.original_modelset.modelsets[1].modelset
    assert type(ns) is beer.NormalSet and ns.cov_type == 'diagonal'
    assert type(ns.means_precisions.posterior) is beer.dists.NormalGamma
    assert ns.means_precisions.posterior.params.mean.shape == (48, 3)
    units, emissions = compat.load(open(os.path.join(GOLDEN, 'ref_units.pkl'), 'rb'))
    assert type(units['a']) is beer.graph.Graph and type(emissions) is beer.JointModelSet
    import sys
    before = {k: v for k, v in sys.modules.items() if k == 'beer' or k.startswith('beer.')}
    alis = compat.load_npz(os.path.join(GOLDEN, 'ref_alis.npz'))
    ali = alis['utt0'][0]
    # (no module aliasing: an installed reference would not be shadowed)
    assert {k: v for k, v in sys.modules.items() if k == 'beer' or k.startswith('beer.')} == before
    with compat.reference_aliases():                     # the legacy route gives the same objects
        legacy = np.load(os.path.join(GOLDEN, 'ref_alis.npz'), allow_pickle=True)['utt0'][0]
    assert type(legacy) is type(ali) and legacy.n_states == ali.n_states
    assert type(ali) is beer.graph.CompiledGraph and ali.n_states == 10
    with pytest.raises(ValueError):                      # same error behaviour as the reference
        beer.evidence_lower_bound(datasize=3) + beer.evidence_lower_bound(datasize=4)


@pytest.mark.gpu
def test_cli_accumulate_update_decode_reproduce_the_reference_run(tmp_path):
    g = load_golden('g13_cli_reference_run')
    feats = os.path.join(GOLDEN, 'ref_feats.npz')
    run(['dataset', 'create', str(tmp_path), feats, str(tmp_path / 'ds.pkl')])
    mdl, alis = os.path.join(GOLDEN, 'ref_phoneloop.pkl'), os.path.join(GOLDEN, 'ref_alis.npz')
    uttids = 'utt0\nutt1\nutt2\n'
    # forced alignment graphs written by the reference
    run(['hmm', 'accumulate', '--alis', alis, mdl, str(tmp_path / 'ds.pkl'),
         str(tmp_path / 'elbo_ali.pkl')], stdin=uttids)
    elbo, count = pickle.load(open(tmp_path / 'elbo_ali.pkl', 'rb'))
    assert count == 3
    # fp32 model: within 1e-5 of the REFERENCE's float64 run of the same inputs (golden
    # g13_cli_reference_run_fp64), or within the error of its own float32 run
    from helpers import assert_within_f32_band
    g64 = load_golden('g13_cli_reference_run_fp64')
    assert_within_f32_band(float(elbo), float(g64['ali_elbo']), g['ali_elbo'], 'aligned ELBO')
    # free phone loop in two shards, reduced by `update`
    run(['hmm', 'accumulate', mdl, str(tmp_path / 'ds.pkl'), str(tmp_path / 'e1.pkl')],
        stdin='utt0\nutt2\n')
    run(['hmm', 'accumulate', mdl, str(tmp_path / 'ds.pkl'), str(tmp_path / 'e2.pkl')],
        stdin='utt1\nmissing_utt\n')
    run(['hmm', 'update', '-o', str(tmp_path / 'optim.pth'), mdl, str(tmp_path / '1.mdl')],
        stdin=f"{tmp_path / 'e1.pkl'}\n{tmp_path / 'e2.pkl'}\n")
    e1, c1 = pickle.load(open(tmp_path / 'e1.pkl', 'rb'))
    e2, c2 = pickle.load(open(tmp_path / 'e2.pkl', 'rb'))
    assert (c1, c2) == (2, 1)
    new = pickle.load(open(tmp_path / '1.mdl', 'rb'))
    # fp64 truth of the same float32 inputs: the same iteration run by the REFERENCE with
    # the model and the features cast to float64 (golden g13_cli_reference_run_fp64,
    # make_golden.py:g13_fp64).  The float32 CLI run must be within 1e-5 of it, or within
    # the error of the reference's own float32 run where float32 cannot do that.
    assert_within_f32_band(float(e1 + e2), float(g64['free_elbo']), g['free_elbo'], 'free-loop ELBO')
    for i, p in enumerate(new.bayesian_parameters()):
        for name, ref, truth in zip(p.posterior._std_params_def,
                                    std_params(g, f'updated.p{i}.posterior'),
                                    std_params(g64, f'updated.p{i}.posterior')):
            got = getattr(p.posterior.params, name).cpu().numpy()
            assert_within_f32_band(got.reshape(ref.shape), truth.reshape(ref.shape), ref,
                                   f'updated p{i}.{name}')
    assert torch.load(tmp_path / 'optim.pth') == {'lrate': 1., 'update_count': 1}
    out = run(['hmm', 'decode', '--per-frame', mdl, str(tmp_path / 'ds.pkl')])
    lines = dict(l.split(' ', 1) for l in out.strip().split('\n'))
    assert sorted(lines) == ['utt0', 'utt1', 'utt2']
    # per-frame phones of the reference's own Viterbi path
    ploop = compat.load(open(mdl, 'rb'))
    from beer_amd.cli.hmm import phones_of_path
    lens = [35, 50, 41]
    off = np.concatenate([[0], np.cumsum(lens)])
    for i, utt in enumerate(sorted(lines)):
        ref_path = g['decode'][off[i]:off[i + 1]].tolist()
        assert lines[utt].split() == phones_of_path(ref_path, ploop.start_pdf, True)
