"""Inputs of the config-5 workflow fixture (shared by the golden generator, which runs the
reference, and tests/test_workflow.py, which runs bin/beer): a synthetic corpus and the
configuration of recipes/aud (conf/mfcc.yml, a conf/hmm.yml of fixture size)."""
import os

import numpy as np

SEED, EPOCHS, SRATE = 1, 3, 16000
PHONES = {'a': (300, 900), 'b': (500, 1500), 'c': (700, 2100), 'd': (1100, 2700)}
UNITS = 'sil non-speech-unit\n' + ''.join(f'{p} speech-unit\n' for p in PHONES)
FEA_CONF = '''srate: 16000
preemph: 0.97
window_len: 0.025
framerate: 0.01
apply_fbank: yes
nfilters: 26
cutoff_hfreq: 7500
cutoff_lfreq: 80
apply_deltas: yes
delta_order: 2
delta_winlen: 2
apply_dct: yes
n_dct_coeff: 12
lifter_coeff: 22
add_energy: yes
utt_mnorm: yes
'''
_TOPO3 = '''  - {start_id: 0, end_id: 1, trans_prob: 1.0}
  - {start_id: 1, end_id: 1, trans_prob: 0.75}
  - {start_id: 1, end_id: 2, trans_prob: 0.25}
  - {start_id: 2, end_id: 2, trans_prob: 0.75}
  - {start_id: 2, end_id: 3, trans_prob: 0.25}
  - {start_id: 3, end_id: 3, trans_prob: 0.75}
  - {start_id: 3, end_id: 4, trans_prob: 0.25}
'''
HMM_CONF = f'''- group_name: non-speech-unit
  n_normal_per_state: 3
  prior_strength: 1.
  noise_std: 0.1
  cov_type: diagonal
  shared_cov: no
  topology:
  - {{start_id: 0, end_id: 1, trans_prob: 1.0}}
  - {{start_id: 1, end_id: 1, trans_prob: 0.75}}
  - {{start_id: 1, end_id: 2, trans_prob: 0.25}}
  - {{start_id: 2, end_id: 2, trans_prob: 0.75}}
  - {{start_id: 2, end_id: 3, trans_prob: 0.25}}
- group_name: speech-unit
  n_normal_per_state: 2
  prior_strength: 1.
  noise_std: 0.1
  cov_type: diagonal
  shared_cov: no
  topology:
{_TOPO3}'''


def synth_corpus(n_utts=20, seed=17):
    '''Utterances of 3..5 "phones" (two sinusoids + noise, 70..130 ms each) between
    two stretches of low noise: int16 audio, transcriptions, utterance ids.'''
    rng = np.random.RandomState(seed)
    names = list(PHONES)
    out, trans, ids = {}, [], []
    for u in range(n_utts):
        seq = ['sil'] + [names[i] for i in rng.randint(0, len(names), rng.randint(3, 6))] + ['sil']
        sig = []
        for p in seq:
            n = int(SRATE * rng.uniform(.07, .13))
            t = np.arange(n) / SRATE
            if p == 'sil':
                x = rng.randn(n) * 30
            else:
                f1, f2 = PHONES[p]
                x = 2000 * np.sin(2 * np.pi * f1 * t) + 1200 * np.sin(2 * np.pi * f2 * t) + \
                    rng.randn(n) * 100
            sig.append(x)
        uid = f'utt{u:02d}'
        out['audio.' + uid] = np.concatenate(sig).astype(np.int16)
        trans.append(uid + ' ' + ' '.join(seq))
        ids.append(uid)
    out['trans'] = np.asarray(trans)
    out['uttids'] = np.asarray(ids)
    return out


def shards(uttids, n=2):
    'The map step\'s split of the utterance list (two accumulate jobs per epoch).'
    return [uttids[i::n] for i in range(n)]


def write_inputs(corpus, root):
    'Lay the corpus and the configuration out as the recipe expects them; returns the paths.'
    from scipy.io import wavfile
    wavdir = os.path.join(root, 'wav')
    os.makedirs(wavdir, exist_ok=True)
    os.makedirs(os.path.join(root, 'feats'), exist_ok=True)
    os.makedirs(os.path.join(root, 'lang'), exist_ok=True)
    paths = {k: os.path.join(root, v) for k, v in dict(
        feaconf='mfcc.yml', hmmconf='hmm.yml', wavscp='wav.scp', trans='trans', feadir='feats',
        feats='feats.npz', dataset='dataset.pkl', units='lang/units', hmms='hmms.mdl',
        ploop_graph='ploop_graph.pkl', decode_graph='decode_graph.pkl', alidir='aligraphs',
        alis='alis.npz').items()}
    lines = []
    for uid in corpus['uttids'].tolist():
        wav = os.path.join(wavdir, uid + '.wav')
        wavfile.write(wav, SRATE, corpus['audio.' + uid])
        lines.append(f'{uid} {wav}')
    open(paths['wavscp'], 'w').write('\n'.join(lines) + '\n')
    open(paths['trans'], 'w').write('\n'.join(corpus['trans'].tolist()) + '\n')
    open(paths['feaconf'], 'w').write(FEA_CONF)
    open(paths['hmmconf'], 'w').write(HMM_CONF)
    open(paths['units'], 'w').write(UNITS)
    return paths
