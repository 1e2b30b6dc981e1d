"""Golden of BASELINE config 5's workflow at fixture size: the
commands of recipes/aud/steps/monophone.sh:62-146 (+ extract_features / create_dataset /
decode) run with the REFERENCE's own command line on a synthetic 20-utterance corpus.

    python tests/golden/make_workflow_golden.py          (build container only)

writes  g17_corpus.npz    the corpus: int16 audio per utterance, transcriptions, units
        g17_workflow.npz  what the reference produced: features, initial models, per-epoch
                          ELBO, final posteriors, decoded phone strings
        g17_workflow_fp64.npz  the same training run by the reference in float64 (model and
                          features cast to double): per-epoch ELBO and final posteriors
Only data is stored.  tests/test_workflow.py replays the same commands through bin/beer.
"""
import os
import pickle
import shutil
import subprocess
import sys
import tempfile
import zipfile

import numpy as np

sys.dont_write_bytecode = True     # (the reference tree is read-only: no __pycache__ into it)

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from workflow_conf import (EPOCHS, FEA_CONF, HMM_CONF, PHONES, SEED, UNITS,   # noqa: E402
                           shards, synth_corpus, write_inputs)

REF = [sys.executable, os.path.join(HERE, 'run_reference_cli.py')]


def ref(args, stdin=None, cwd=None):
    p = subprocess.run(REF + args, input=stdin, capture_output=True, text=True, cwd=cwd,
                       env=dict(os.environ, PYTHONDONTWRITEBYTECODE='1'))
    if p.returncode != 0:
        raise RuntimeError(f'reference beer {args}: {p.stderr[-2000:]}')
    return p.stdout


STD_NAMES = {
    'NormalWishart': ('mean', 'scale', 'scale_matrix', 'dof'),
    'NormalGamma': ('mean', 'scale', 'shape', 'rates'),
    'IsotropicNormalGamma': ('mean', 'scale', 'shape', 'rate'),
    'Dirichlet': ('concentrations',),
    'Gamma': ('shape', 'rate'),
}


def params_of(model):
    'Posterior standard parameters of every Bayesian parameter, in model order.'
    out = {}
    for i, param in enumerate(model.bayesian_parameters()):
        post = param.posterior
        for n in STD_NAMES[type(post).__name__]:
            out[f'p{i}.{n}'] = getattr(post.params, n).detach().cpu().numpy()
    return out


def fp64_run(beer, mdl0, paths, uttids):
    '''The same EPOCHS of aligned training run by the REFERENCE in float64 -- initial model,
    features cast to double, the alignment graphs as `mkaligraph` wrote them --
    through the loop of accumulate.py:39-59 / update.py:41-62 (one shard: the sum over
    utterances does not depend on how they are dealt to jobs): the fp64 truth of the same
    float32 inputs for tests/test_workflow.py, so that the test does not take it from
    the build's own fp64 path (as g13_cli_reference_run_fp64 does for the CLI replay).'''
    import torch
    model = pickle.load(open(mdl0, 'rb')).double()
    dataset = pickle.load(open(paths['dataset'], 'rb'))
    alis = np.load(paths['alis'], allow_pickle=True)
    optim = beer.VBConjugateOptimizer(model.conjugate_bayesian_parameters(keepgroups=True), 1.)
    logged = []
    for _ in range(EPOCHS):
        optim.init_step()
        elbo = beer.evidence_lower_bound(datasize=dataset.size)
        count = 0
        for uttid in uttids:
            utt = dataset[uttid]
            elbo += beer.evidence_lower_bound(model, utt.features.double(),
                                              inference_graph=alis[uttid][0],
                                              datasize=dataset.size, scale=1.)
            count += 1
        logged.append(float(elbo) / (count * dataset.size))
        elbo.backward()
        optim.step()
    out = {'logged_elbo': np.asarray(logged)}
    for k, v in params_of(model).items():
        out['final.' + k] = v
    print('fp64 logged ELBO per epoch:', logged)
    return out


def main():
    corpus = synth_corpus()
    np.savez_compressed(os.path.join(HERE, 'g17_corpus.npz'), **corpus)
    tmp = tempfile.mkdtemp(prefix='g17_')
    try:
        paths = write_inputs(corpus, tmp)
        out = {}
        ref(['features', 'extract', paths['feaconf'], paths['wavscp'], paths['feadir']])
        ref(['features', 'archive', paths['feadir'], paths['feats']])
        ref(['dataset', 'create', tmp, paths['feats'], paths['dataset']])
        arch = np.load(paths['feats'])
        for utt in ('utt00', 'utt07'):
            out[f'feats.{utt}'] = arch[utt]
        out['n_frames'] = np.asarray(sum(len(arch[u]) for u in arch.files))
        ref(['-s', str(SEED), 'hmm', 'mkphones', '-d', paths['dataset'], paths['hmmconf'],
             paths['units'], paths['hmms']])
        ref(['hmm', 'mkphoneloopgraph', '--start-end-group', 'non-speech-unit', paths['units'],
             paths['ploop_graph']])
        ref(['hmm', 'mkdecodegraph', paths['ploop_graph'], paths['hmms'], paths['decode_graph']])
        ref(['hmm', 'mkphoneloop', '--weights-prior', 'gamma_dirichlet_process',
             paths['decode_graph'], paths['hmms'], os.path.join(tmp, '0.mdl')])
        os.makedirs(paths['alidir'])
        ref(['hmm', 'mkaligraph', paths['hmms'], paths['alidir']], stdin=open(paths['trans']).read())
        with zipfile.ZipFile(paths['alis'], 'w') as z:
            for f in sorted(os.listdir(paths['alidir'])):
                z.write(os.path.join(paths['alidir'], f), f)
        sys.path.insert(0, '/root/reference')
        import types
        sys.modules.setdefault('natsort', types.SimpleNamespace(natsorted=sorted))
        import beer                                      # noqa: F401  (to unpickle)
        model0 = pickle.load(open(os.path.join(tmp, '0.mdl'), 'rb'))
        for k, v in params_of(model0).items():
            out['init.' + k] = v
        out['init.trans'] = model0.graph.trans_log_probs.numpy()
        out['init.pdf_id_mapping'] = np.asarray(model0.graph.pdf_id_mapping)
        uttids = sorted(corpus['uttids'].tolist())
        logged = []
        mdl = os.path.join(tmp, '0.mdl')
        for epoch in range(1, EPOCHS + 1):
            pkls = []
            for j, shard in enumerate(shards(uttids)):
                pkl = os.path.join(tmp, f'elbo_{epoch}_{j}.pkl')
                ref(['hmm', 'accumulate', '--alis', paths['alis'], mdl, paths['dataset'], pkl],
                    stdin='\n'.join(shard) + '\n')
                pkls.append(pkl)
            new = os.path.join(tmp, f'{epoch}.mdl')
            ref(['hmm', 'update', '-o', os.path.join(tmp, 'optim.pth'), mdl, new],
                stdin='\n'.join(pkls) + '\n')
            total, count = None, 0
            for pkl in pkls:
                e, c = pickle.load(open(pkl, 'rb'))
                total, count = (e if total is None else total + e), count + c
            logged.append(float(total) / (count * total._datasize))
            mdl = new
        out['logged_elbo'] = np.asarray(logged)
        final = pickle.load(open(mdl, 'rb'))
        for k, v in params_of(final).items():
            out['final.' + k] = v
        out['final.trans'] = final.graph.trans_log_probs.numpy()
        dec = ref(['hmm', 'decode', mdl, paths['dataset']])
        out['decode'] = np.asarray(sorted(l for l in dec.strip().split('\n') if l))
        dec0 = ref(['hmm', 'decode', os.path.join(tmp, '0.mdl'), paths['dataset']])
        out['decode_init'] = np.asarray(sorted(l for l in dec0.strip().split('\n') if l))
        out['phonelist'] = np.asarray(ref(['hmm', 'phonelist', paths['hmms']]).split())
        np.savez_compressed(os.path.join(HERE, 'g17_workflow.npz'), **out)
        np.savez_compressed(os.path.join(HERE, 'g17_workflow_fp64.npz'),
                            **fp64_run(beer, os.path.join(tmp, '0.mdl'), paths, uttids))
        print('logged ELBO per epoch:', logged)
        print('\n'.join(out['decode'][:4]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
