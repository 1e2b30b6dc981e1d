"""CPU oracle for the feature front-end (beer/features.py and the pipeline of
beer/cli/subcommands/features/extract.py).

TEST INFRASTRUCTURE ONLY: imported by `tests/` (and nothing in `beer_amd`).

Parity status: PINNED against the reference's own golden files
(`tests/audio.npy`, `tests/fbank.npy`, `tests/fbank_d_dd.npy` of the reference,
kept as `tests/golden/ref_audio.npy`, `ref_fbank.npy`, `ref_fbank_d_dd.npy`:
beer/tests/test_features.py:17-28) and against outputs of the reference's
functions on the same audio for the command-line pipeline
(`tests/golden/g15_features.npz`, made by tests/golden/make_golden.py:g15).

numpy, float64, written independently of the reference's code: frames are
gathered with an index matrix, the derivative filter is an explicit sum.
"""

import math

import numpy as np


def hz2mel(f):
    'features.py:11-13'
    return 1127 * np.log(1 + f / 700.0)


def mel2hz(m):
    'features.py:16-18'
    return 700.0 * (np.exp(m / 1127.0) - 1)


def triangular_filters(nfilters, fft_len=512, srate=16000, lowfreq=0, highfreq=None,
                       align=True):
    '''features.py:31-83.  Filter i rises on bins [e_i, e_{i+1}] and falls on
    [e_{i+1}, e_{i+2}]; each side is `linspace` between its end-point values.'''
    highfreq = highfreq or srate / 2
    e = fft_len * mel2hz(np.linspace(hz2mel(lowfreq), hz2mel(highfreq), nfilters + 2)) / srate
    if align:
        e = np.floor(e)
    bins = np.arange(fft_len // 2)
    F = np.zeros((nfilters, fft_len // 2))
    for i in range(nfilters):
        a, c, b = e[i], e[i + 1], e[i + 2]
        up = np.flatnonzero((bins >= a) & (bins <= c))
        F[i, up] = np.linspace((1. / (c - a)) * (up[0] - a),
                               (1. / (c - a)) * (up[-1] - a), len(up))
        dn = np.flatnonzero((bins >= c) & (bins <= b))
        F[i, dn] = np.linspace((1. / (b - c)) * (b - dn[0]), (1. / (b - c)) * (b - dn[-1]),
                               len(dn))
    return F


def _geometry(n, srate, flen, frate):
    step, size = int(srate * frate), int(srate * flen)
    nframes = (n - size) // step + 1
    fft_len = int(2 ** np.floor(np.log2(size) + 1))
    index = np.arange(nframes)[:, None] * step + np.arange(size)[None, :]
    return step, size, nframes, fft_len, index


def fbank(signal, flen=0.025, frate=0.01, hifreq=8000, lowfreq=20, nfilters=26,
          preemph=0.97, srate=16000):
    '''features.py:148-204: float32 pre-emphasis of the whole signal, Hamming
    window, |rfft| without the Nyquist bin, triangular filters, log(1 + .).'''
    _, size, _, fft_len, index = _geometry(len(signal), srate, flen, frate)
    x = np.asarray(signal).astype(np.float32)
    prev = np.concatenate([x[:1], x[:-1]])
    x = x - np.float32(preemph) * prev                       # float32 arithmetic
    frames = x[index].astype(np.float64) * np.hamming(size)[None, :]
    mag = np.abs(np.fft.rfft(frames, n=fft_len, axis=-1))[:, :fft_len // 2]
    F = triangular_filters(nfilters, fft_len, srate, lowfreq, hifreq)
    return np.log(mag @ F.T + 1)


def short_term_mspec(signal, flen=0.025, frate=0.01, preemph=0.97, srate=16000):
    '''features.py:107-146: remove the DC offset, pre-emphasis inside every
    frame (its first sample is filtered with itself), Hamming window, |rfft|.'''
    _, size, _, fft_len, index = _geometry(len(signal), srate, flen, frate)
    x = np.asarray(signal) - np.asarray(signal).mean()
    frames = x[index].astype(np.float64)
    shifted = np.concatenate([frames[:, :1], frames[:, :-1]], axis=1)
    frames = (frames - preemph * shifted) * np.hamming(size)[None, :]
    return np.abs(np.fft.rfft(frames, n=fft_len, axis=-1))[:, :fft_len // 2], fft_len


def add_deltas(fea, winlens=(2, 2)):
    '''features.py:86-105: each order is the FIR filter j / (2 sum_j j^2),
    j = -w..w, over the previous block with replicated edges.'''
    blocks = [np.asarray(fea, dtype=np.float64)]
    for w in winlens:
        cur = blocks[-1]
        T = len(cur)
        den = 2.0 * sum(j * j for j in range(-w, w + 1))
        out = np.zeros_like(cur)
        for j in range(-w, w + 1):
            out += (j / den) * cur[np.clip(np.arange(T) + j, 0, T - 1)]
        blocks.append(out)
    return np.concatenate(blocks, axis=1)


DEFAULT_CONF = {
    'srate': 16000, 'preemph': 0.97, 'window_len': 0.025, 'framerate': 0.01,
    'apply_fbank': True, 'nfilters': 26, 'cutoff_hfreq': 8000, 'cutoff_lfreq': 20,
    'apply_deltas': True, 'delta_order': 2, 'delta_winlen': 2, 'apply_dct': True,
    'n_dct_coeff': 13, 'lifter_coeff': 22, 'utt_mnorm': False, 'add_energy': True,
}


def extract(signal, conf=None):
    'extract.py:107-161 for one utterance.'
    c = dict(DEFAULT_CONF)
    c.update(conf or {})
    spec, fft_len = short_term_mspec(signal, c['window_len'], c['framerate'], c['preemph'],
                                     c['srate'])
    nf = c['nfilters']
    if c['apply_fbank']:
        spec = spec @ triangular_filters(nf, fft_len, 16000, c['cutoff_lfreq'],
                                         c['cutoff_hfreq']).T
    logspec = np.log(1e-6 + spec)
    norm = math.sqrt(2. / nf)
    if c['apply_dct']:
        ncoef = c['n_dct_coeff']
        basis = np.cos(np.pi / nf * np.outer(np.arange(nf) + .5, np.arange(1, ncoef + 1)))
        L = c['lifter_coeff']
        fea = (logspec @ basis) * norm * (1 + (L / 2) * np.sin(np.pi * np.arange(1, ncoef + 1) / L))
    else:
        fea = logspec
    if c['add_energy']:
        fea = np.concatenate([(logspec.sum(axis=-1) * norm)[:, None], fea], axis=1)
    if c['apply_deltas']:
        fea = add_deltas(fea, (c['delta_winlen'],) * c['delta_order'])
    if c['utt_mnorm']:
        fea = fea - fea.mean(axis=0, keepdims=True)
    return fea
