"""Feature front-end: the numpy oracle against
the reference's data files and outputs (CPU), and the HIP kernels against the
same goldens (GPU)."""

import os
import sys

import numpy as np
import pytest

from helpers import load_golden

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import features_oracle as fo          # noqa: E402

GOLD = os.path.join(ROOT, 'tests', 'golden')
SIGNALS = ('audio', 'synth', 'synthf')
CMN_CONF = {'apply_dct': False, 'utt_mnorm': True, 'nfilters': 40, 'add_energy': False,
            'delta_order': 1, 'delta_winlen': 3}


def _signal(g, name):
    if name == 'audio':
        return np.load(os.path.join(GOLD, 'ref_audio.npy'))
    return g['synth'] if name == 'synth' else g['synth'] / 32768.


def _close(a, b, tol=1e-9):
    assert a.shape == b.shape
    scale = max(1., float(np.abs(b).max()))
    assert float(np.abs(a - b).max()) <= tol * scale, float(np.abs(a - b).max())


# ---- oracle (CPU) ------------------------------------------------------------

def test_oracle_deltas_against_reference_data_files():
    'tests/fbank.npy -> tests/fbank_d_dd.npy of the reference (test_features.py:23-28).'
    fea = np.load(os.path.join(GOLD, 'ref_fbank.npy'))
    ref = np.load(os.path.join(GOLD, 'ref_fbank_d_dd.npy'))
    assert np.allclose(fo.add_deltas(fea), ref, rtol=1e-12, atol=1e-12)


def test_reference_fbank_file_is_stale():
    '''The reference's fbank.npy does not come from its current fbank() (its own
    test_features.py fails, max |diff| = 0.32): the oracle is pinned on
    outputs of the reference's functions instead (g15).'''
    g = load_golden('g15_features')
    stale = np.load(os.path.join(GOLD, 'ref_fbank.npy'))
    assert np.abs(g['audio.fbank30'] - stale).max() > .1


@pytest.mark.parametrize('name', SIGNALS)
def test_oracle_against_reference_outputs(name):
    g = load_golden('g15_features')
    sig = _signal(g, name)
    _close(fo.fbank(sig, nfilters=30, lowfreq=100), g[f'{name}.fbank30'], 1e-12)
    _close(fo.fbank(sig), g[f'{name}.fbank26'], 1e-12)
    _close(fo.short_term_mspec(sig)[0], g[f'{name}.mspec'], 1e-12)
    _close(fo.extract(sig), g[f'{name}.mfcc'], 1e-11)
    _close(fo.extract(sig, CMN_CONF), g[f'{name}.fbank_cmn'], 1e-11)
    _close(fo.triangular_filters(30, 512, 16000, 100, 8000), g['filters30'], 1e-15)


def test_host_tables_match_reference():
    import beer_amd as beer
    g = load_golden('g15_features')
    assert np.array_equal(beer.features.create_fbank(30, 512, lowfreq=100, highfreq=8000),
                          g['filters30'])
    assert abs(beer.features.hz2mel(1000.) - 1127 * np.log(1 + 1000 / 700.)) < 1e-12
    assert abs(beer.features.mel2hz(beer.features.hz2mel(440.)) - 440.) < 1e-9
    assert abs(beer.features.bark2hz(beer.features.hz2bark(440.)) - 440.) < 1e-9


# ---- HIP kernels (GPU) -------------------------------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize('name', SIGNALS)
def test_gpu_features_against_reference(name):
    import beer_amd as beer
    g = load_golden('g15_features')
    sig = _signal(g, name)
    _close(beer.features.fbank(sig, nfilters=30, lowfreq=100), g[f'{name}.fbank30'])
    _close(beer.features.fbank(sig), g[f'{name}.fbank26'])
    spec, fft_len = beer.features.short_term_mspec(sig)
    assert fft_len == 512
    _close(spec, g[f'{name}.mspec'])
    _close(beer.features.extract([sig])[0], g[f'{name}.mfcc'])
    _close(beer.features.extract([sig], CMN_CONF)[0], g[f'{name}.fbank_cmn'])


@pytest.mark.gpu
def test_gpu_deltas_against_reference_data_files():
    import beer_amd as beer
    fea = np.load(os.path.join(GOLD, 'ref_fbank.npy'))
    ref = np.load(os.path.join(GOLD, 'ref_fbank_d_dd.npy'))
    assert np.allclose(beer.features.add_deltas(fea), ref, rtol=1e-12, atol=1e-12)


@pytest.mark.gpu
def test_gpu_ragged_batch_equals_per_utterance():
    'A batch of utterances of different lengths == one call per utterance.'
    import beer_amd as beer
    rng = np.random.RandomState(0)
    sigs = [(2000 * rng.randn(n)).astype(np.int16) for n in (400, 3217, 1600, 16000, 801)]
    for conf in ({}, CMN_CONF):
        batch = beer.features.extract(sigs, conf)
        for sig, fea in zip(sigs, batch):
            single = beer.features.extract([sig], conf)[0]
            assert fea.shape == single.shape == fo.extract(sig, conf).shape
            assert np.array_equal(fea, single)
            _close(fea, fo.extract(sig, conf))


@pytest.mark.gpu
def test_gpu_features_edge_cases():
    import beer_amd as beer
    # a signal shorter than one frame yields no frame; an empty list no output
    short = np.zeros(100, dtype=np.int16)
    out = beer.features.extract([short, np.ones(400, dtype=np.int16)])
    assert out[0].shape == (0, 42) and out[1].shape == (1, 42)
    assert beer.features.extract([]) == []
    # other window lengths / FFT sizes: 10 ms frames (fft 256), 50 ms (fft 1024)
    rng = np.random.RandomState(1)
    sig = 1000 * rng.randn(5000)
    for wl in (0.010, 0.050):
        conf = {'window_len': wl, 'apply_deltas': False}
        _close(beer.features.extract([sig], conf)[0], fo.extract(sig, conf))


@pytest.mark.gpu
def test_cli_features_extract_archive_dataset(tmp_path):
    '`beer features extract` -> `features archive` -> `dataset create` on WAV files.'
    import pickle
    from scipy.io import wavfile
    from beer_amd.cli import main as cli_main
    g = load_golden('g15_features')
    wavs = {'utt_audio': _signal(g, 'audio'), 'utt_synth': _signal(g, 'synth')}
    lines = []
    for name, sig in wavs.items():
        path = tmp_path / f'{name}.wav'
        wavfile.write(str(path), 16000, sig)
        # the second one through a shell command, as Kaldi-style lists do
        lines.append(f'{name} {path}' if name == 'utt_audio' else f'{name} cat {path} |')
    (tmp_path / 'wavs.scp').write_text('\n'.join(lines) + '\n')
    (tmp_path / 'fea.yml').write_text('nfilters: 26\n')
    feadir = tmp_path / 'fea'
    feadir.mkdir()
    cli_main.main(['features', 'extract', str(tmp_path / 'fea.yml'), str(tmp_path / 'wavs.scp'),
                   str(feadir)])
    _close(np.load(feadir / 'utt_audio.npy'), g['audio.mfcc'])
    _close(np.load(feadir / 'utt_synth.npy'), g['synth.mfcc'])
    cli_main.main(['features', 'archive', str(feadir), str(tmp_path / 'fea.npz')])
    arch = np.load(tmp_path / 'fea.npz')
    assert sorted(arch.files) == ['utt_audio', 'utt_synth']
    cli_main.main(['dataset', 'create', str(tmp_path), str(tmp_path / 'fea.npz'),
                   str(tmp_path / 'data.pkl')])
    with open(tmp_path / 'data.pkl', 'rb') as f:
        dataset = pickle.load(f)
    assert dataset.size == len(g['audio.mfcc']) + len(g['synth.mfcc'])
    with pytest.raises(SystemExit):
        (tmp_path / 'bad.yml').write_text('no_such_option: 1\n')
        cli_main.main(['features', 'extract', str(tmp_path / 'bad.yml'),
                       str(tmp_path / 'wavs.scp'), str(feadir)])
