"""BASELINE config 5 at fixture size: the recipe's workflow -- features extract -> archive ->
dataset create -> mkphones / mkphoneloopgraph / mkdecodegraph / mkphoneloop -> mkaligraph ->
N epochs of (accumulate in two shards + update) -> decode -- run through `bin/beer` on a
synthetic 20-utterance corpus, against the same commands run with the REFERENCE's own
command line in the build container (tests/golden/make_workflow_golden.py -> g17_*.npz;
flow of recipes/aud/steps/monophone.sh:62-146, extract_features.sh, create_dataset.sh,
decode.sh)."""

import io
import logging
import os
import pickle
import sys
import zipfile

import numpy as np
import pytest

from helpers import GOLDEN, assert_close, load_golden

sys.path.insert(0, GOLDEN)
from workflow_conf import EPOCHS, SEED, shards, write_inputs              # noqa: E402

from beer_amd.cli import main as cli_main                                 # noqa: E402


def run(argv, stdin=''):
    old_in, old_out = sys.stdin, sys.stdout
    sys.stdin, sys.stdout = io.StringIO(stdin), io.StringIO()
    try:
        cli_main.main(argv)
        return sys.stdout.getvalue()
    finally:
        sys.stdin, sys.stdout = old_in, old_out


def posterior_arrays(model):
    out = {}
    for i, param in enumerate(model.bayesian_parameters()):
        post = param.posterior
        for n in post._std_params_def:
            out[f'p{i}.{n}'] = getattr(post.params, n).detach().cpu().numpy()
    return out


@pytest.mark.gpu
def test_config5_workflow_reproduces_the_reference_run(tmp_path):
    corpus, g = load_golden('g17_corpus'), load_golden('g17_workflow')
    tmp = str(tmp_path)
    paths = write_inputs(corpus, tmp)
    # ---- front end: float64 features, bit-level agreement is not expected (FFT order),
    # 1e-9 is
    run(['features', 'extract', paths['feaconf'], paths['wavscp'], paths['feadir']])
    run(['features', 'archive', paths['feadir'], paths['feats']])
    run(['dataset', 'create', tmp, paths['feats'], paths['dataset']])
    arch = np.load(paths['feats'])
    assert sorted(arch.files) == sorted(corpus['uttids'].tolist())
    for utt in ('utt00', 'utt07'):
        assert_close(arch[utt], g[f'feats.{utt}'], 1e-9, 'features ' + utt)
    assert sum(len(arch[u]) for u in arch.files) == int(g['n_frames'])
    # ---- model building: the same seed gives the same initial model (the initial noise
    # is drawn by the same torch calls in the same order)
    run(['-s', str(SEED), 'hmm', 'mkphones', '-d', paths['dataset'], paths['hmmconf'],
         paths['units'], paths['hmms']])
    run(['hmm', 'mkphoneloopgraph', '--start-end-group', 'non-speech-unit', paths['units'],
         paths['ploop_graph']])
    run(['hmm', 'mkdecodegraph', paths['ploop_graph'], paths['hmms'], paths['decode_graph']])
    mdl = os.path.join(tmp, '0.mdl')
    run(['hmm', 'mkphoneloop', '--weights-prior', 'gamma_dirichlet_process',
         paths['decode_graph'], paths['hmms'], mdl])
    assert run(['hmm', 'phonelist', paths['hmms']]).split() == g['phonelist'].tolist()
    model0 = pickle.load(open(mdl, 'rb'))
    for k, v in posterior_arrays(model0).items():
        ref = g['init.' + k]
        assert_close(v.reshape(ref.shape), ref, 1e-6, 'initial ' + k)
    assert model0.graph.pdf_id_mapping == g['init.pdf_id_mapping'].tolist()
    assert_close(np.exp(model0.graph.trans_log_probs.cpu().numpy()), np.exp(g['init.trans']), 1e-6,
                 'initial transitions')
    os.makedirs(paths['alidir'])
    run(['hmm', 'mkaligraph', paths['hmms'], paths['alidir']], stdin=open(paths['trans']).read())
    with zipfile.ZipFile(paths['alis'], 'w') as z:
        for f in sorted(os.listdir(paths['alidir'])):
            z.write(os.path.join(paths['alidir'], f), f)
    # the untrained model decodes what the reference's untrained model decodes
    dec0 = run(['hmm', 'decode', mdl, paths['dataset']])
    assert sorted(l for l in dec0.strip().split('\n') if l) == g['decode_init'].tolist()
    # ---- training: EPOCHS x (two accumulate jobs + update), float32 as the reference's CLI
    uttids = sorted(corpus['uttids'].tolist())
    logged = []
    for epoch in range(1, EPOCHS + 1):
        pkls = []
        for j, shard in enumerate(shards(uttids)):
            pkl = os.path.join(tmp, f'elbo_{epoch}_{j}.pkl')
            run(['hmm', 'accumulate', '--alis', paths['alis'], mdl, paths['dataset'], pkl],
                stdin='\n'.join(shard) + '\n')
            pkls.append(pkl)
        new = os.path.join(tmp, f'{epoch}.mdl')
        run(['hmm', 'update', '-o', os.path.join(tmp, 'optim.pth'), mdl, new],
            stdin='\n'.join(pkls) + '\n')
        total, count = None, 0
        for pkl in pkls:
            e, c = pickle.load(open(pkl, 'rb'))
            total, count = (e if total is None else total + e), count + c
        logged.append(float(total) / (count * total._datasize))
        mdl = new
    # fp64 truth of the same float32 inputs: the same three epochs run by the REFERENCE with
    # the initial model and the features cast to float64 (golden g17_workflow_fp64,
    # make_workflow_golden.py:fp64_run -- not this build's own fp64 path).  The float32 CLI
    # run must be within 1e-5 of it, or within the error of the reference's own float32 run
    # (`g`) where float32 arithmetic cannot do that: float32 error compounds over three
    # epochs, for the reference as for this build.
    from helpers import assert_within_f32_band
    g64 = load_golden('g17_workflow_fp64')
    for epoch, (got, ref, truth) in enumerate(zip(logged, g['logged_elbo'], g64['logged_elbo']),
                                              start=1):
        assert_within_f32_band(got, truth, ref, f'logged ELBO, epoch {epoch}')
    final = pickle.load(open(mdl, 'rb'))
    for k, v in posterior_arrays(final).items():
        ref, truth = g['final.' + k], g64['final.' + k]
        assert_within_f32_band(v.reshape(ref.shape), truth.reshape(ref.shape), ref, 'final ' + k)
    # ---- decoding with the trained model: the reference's phone strings
    dec = run(['hmm', 'decode', mdl, paths['dataset']])
    assert sorted(l for l in dec.strip().split('\n') if l) == g['decode'].tolist()
