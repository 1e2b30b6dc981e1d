"""Speech feature extraction on the GPU.

API mirror of beer/features.py (`hz2mel`, `mel2hz`, `hz2bark`, `bark2hz`,
`create_fbank`, `add_deltas`, `short_term_mspec`, `fbank`) plus `extract`, the
pipeline of `beer features extract` (beer/cli/subcommands/features/
extract.py:107-161) over a whole list of utterances in a handful of kernel
launches.  The per-utterance functions return numpy arrays like the
reference; the arithmetic runs in `csrc/features.hip` (float64, as numpy's).

The small constant tables (window, triangular filters, cosine bases, lifter)
are built on the host once per configuration, as the reference builds them.
"""

import ctypes
import functools
import math

import numpy as np
import torch

from . import _hip

__all__ = ['hz2mel', 'mel2hz', 'hz2bark', 'bark2hz', 'create_fbank', 'add_deltas',
           'short_term_mspec', 'fbank', 'extract', 'FEACONF', 'dct_bases', 'lifter']


def hz2mel(freq_hz):
    'Hertz -> Mel.'
    return 1127 * np.log(1 + freq_hz / 700.0)


def mel2hz(mel):
    'Mel -> Hertz.'
    return 700.0 * (np.exp(mel / 1127.0) - 1)


def hz2bark(freq_hz):
    'Hertz -> Bark.'
    return (29.81 * freq_hz) / (1960 + freq_hz) - 0.53


def bark2hz(bark):
    'Bark -> Hertz.'
    return (1960 * (bark + .53)) / (29.81 - bark - .53)


@functools.lru_cache(maxsize=8)
def create_fbank(nfilters, fft_len=512, srate=16000, lowfreq=0, highfreq=None,
                 hz2scale=hz2mel, scale2hz=mel2hz, align_filt_center=True):
    '''[nfilters, fft_len // 2] matrix of triangular filters equally spaced on
    a perceptual scale (features.py:47-83; each side of a triangle is a
    linspace between its values on the first and last FFT bin it covers).'''
    highfreq = highfreq or srate / 2
    edges = np.linspace(hz2scale(lowfreq), hz2scale(highfreq), nfilters + 2)
    edges = fft_len * scale2hz(edges) / srate
    if align_filt_center:
        edges = np.floor(edges)
    bins = np.arange(0, fft_len // 2)
    filters = np.zeros((nfilters, fft_len // 2))
    for i in range(nfilters):
        start, center, end = edges[i], edges[i + 1], edges[i + 2]
        up, down = 1. / (center - start), 1. / (end - center)
        row = np.zeros(len(bins))
        rising = (bins >= start) & (bins <= center)
        sel = bins[rising]
        row[rising] = np.linspace(up * (sel[0] - start), up * (sel[-1] - start), len(sel))
        falling = (bins >= center) & (bins <= end)
        sel = bins[falling]
        row[falling] = np.linspace(down * (end - sel[0]), down * (end - sel[-1]), len(sel))
        filters[i] = row
    return filters


def dct_bases(nfilters, n_dct_coeff):
    'Cosine bases [nfilters, n_coeff] of the cepstral transform (extract.py:36-40).'
    m = np.arange(1, n_dct_coeff + 1)[None, :]
    return np.cos(m * np.pi / nfilters * (np.arange(nfilters)[:, None] + 0.5))


def lifter(n_dct_coeff, l_coeff):
    'Cepstral lifter (extract.py:141-144).'
    return 1 + (l_coeff / 2) * np.sin(np.pi * (1 + np.arange(n_dct_coeff)) / l_coeff)


# default configuration of `beer features extract` (extract.py:16-33)
FEACONF = {
    'srate': 16000, 'preemph': 0.97, 'window_len': 0.025, 'framerate': 0.01,
    'apply_fbank': True, 'nfilters': 26, 'cutoff_hfreq': 8000, 'cutoff_lfreq': 20,
    'apply_deltas': True, 'delta_order': 2, 'delta_winlen': 2, 'apply_dct': True,
    'n_dct_coeff': 13, 'lifter_coeff': 22, 'utt_mnorm': False, 'add_energy': True,
}


def _fft_len(flen_samp):
    return int(2 ** np.floor(np.log2(flen_samp) + 1))


def _dev(array, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(array))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(_hip.require_device())


def _in_code(dtype):
    code = {torch.int16: _hip.I16, torch.float32: _hip.F32, torch.float64: _hip.F64}.get(dtype)
    if code is None:
        raise TypeError(f'audio samples must be int16, float32 or float64, got {dtype}')
    return code


class _Batch:
    'Signals of a list of utterances packed back to back on the GPU.'

    def __init__(self, signals, flen, fstep):
        dev = _hip.require_device()
        sigs = []
        for s in signals:
            t = s if isinstance(s, torch.Tensor) else torch.from_numpy(np.array(s))
            if t.dim() != 1:
                raise ValueError('expected one-dimensional (mono) signals')
            if t.dtype not in (torch.int16, torch.float32, torch.float64):
                t = t.to(torch.float64)
            sigs.append(t)
        dtypes = {t.dtype for t in sigs}
        if len(dtypes) > 1:
            sigs = [t.to(torch.float64) for t in sigs]
        self.lengths = [len(t) for t in sigs]
        self.nframes = [max(0, (n - flen) // fstep + 1) for n in self.lengths]
        self.signal = torch.cat([t.to(dev) for t in sigs]) if sigs else \
            torch.zeros(0, dtype=torch.float64, device=dev)
        self.sample_off = torch.tensor(np.concatenate([[0], np.cumsum(self.lengths)]),
                                       dtype=torch.int64).to(dev)
        self.frame_off = torch.tensor(np.concatenate([[0], np.cumsum(self.nframes)]),
                                      dtype=torch.int64).to(dev)
        self.total = int(sum(self.nframes))
        self.nutt = len(sigs)
        self.code = _in_code(self.signal.dtype)


def _run(batch, mode, flen, fstep, preemph, window, filters=None, apply_log=False,
         log_offset=1., dct=None, lift=None, norm=1., add_energy=False, extra_cols=0):
    'Launch beer_features_extract; returns the [total_frames, width] device buffer.'
    fft_len = _fft_len(flen)
    conf = _hip.FeaConf()
    keep = [_dev(window, torch.float64)]
    conf.flen, conf.fstep, conf.fft_len, conf.mode = flen, fstep, fft_len, mode
    conf.preemph, conf.log_offset, conf.norm = preemph, log_offset, norm
    conf.apply_log, conf.add_energy = int(apply_log), int(add_energy)
    conf.window = keep[0].data_ptr()
    nf = fft_len // 2
    if filters is not None:
        nf = filters.shape[0]
        nz = filters != 0
        lo = np.where(nz.any(1), nz.argmax(1), 0).astype(np.int32)
        hi = np.where(nz.any(1), filters.shape[1] - 1 - nz[:, ::-1].argmax(1), -1).astype(np.int32)
        keep += [_dev(filters, torch.float64), _dev(lo), _dev(hi)]
        conf.nfilters = nf
        conf.filters, conf.filt_lo, conf.filt_hi = (k.data_ptr() for k in keep[-3:])
    width = nf
    if dct is not None:
        keep.append(_dev(dct, torch.float64))
        conf.n_dct, conf.dct = dct.shape[1], keep[-1].data_ptr()
        width = dct.shape[1]
        if lift is not None:
            keep.append(_dev(lift, torch.float64))
            conf.lifter = keep[-1].data_ptr()
    base = width + int(add_energy)
    ld = base * (1 + extra_cols)
    out = torch.empty(batch.total, ld, dtype=torch.float64, device=batch.signal.device)
    mean = None
    if mode == 1:
        mean = torch.empty(batch.nutt, dtype=torch.float64, device=out.device)
        _hip.call('beer_features_signal_mean', batch.code, batch.nutt, _hip.ptr(batch.sample_off),
                  _hip.ptr(batch.signal), _hip.ptr(mean))
    _hip.call('beer_features_extract', batch.code, batch.nutt, _hip.ptr(batch.sample_off),
              _hip.ptr(batch.frame_off), batch.total, _hip.ptr(batch.signal), _hip.ptr(mean),
              ctypes.byref(conf), _hip.ptr(out), ld)
    torch.cuda.current_stream().synchronize()          # `keep` tables may go now
    return out, base


def _deltas(buf, frame_off, nutt, base, winlens):
    'Fill the column blocks after the first `base` columns with derivatives.'
    T, ld = buf.shape
    esz = buf.element_size()
    for order, wlen in enumerate(winlens, start=1):
        src = ctypes.c_void_p(buf.data_ptr() + (order - 1) * base * esz)
        dst = ctypes.c_void_p(buf.data_ptr() + order * base * esz)
        _hip.call('beer_features_deltas', nutt, _hip.ptr(frame_off), T, base, ld, int(wlen),
                  src, dst)


def add_deltas(fea, winlens=(2, 2)):
    'Append derivatives (deltas, double deltas, ...) to a feature matrix.'
    fea = np.asarray(fea, dtype=np.float64)
    T, D = fea.shape
    dev = _hip.require_device()
    buf = torch.empty(T, D * (1 + len(winlens)), dtype=torch.float64, device=dev)
    buf[:, :D] = torch.from_numpy(np.ascontiguousarray(fea)).to(dev)
    off = torch.tensor([0, T], dtype=torch.int64, device=dev)
    _deltas(buf, off, 1, D, winlens)
    return buf.cpu().numpy()


def short_term_mspec(signal, flen=0.025, frate=0.01, preemph=0.97, srate=16000,
                     window=np.hamming):
    '(magnitude spectrum [n_frames, fft_len // 2], fft_len).'
    fstep, flen_samp = int(srate * frate), int(srate * flen)
    batch = _Batch([signal], flen_samp, fstep)
    out, _ = _run(batch, 1, flen_samp, fstep, preemph, window(flen_samp))
    return out.cpu().numpy(), _fft_len(flen_samp)


def fbank(signal, flen=0.025, frate=0.01, hifreq=8000, lowfreq=20, nfilters=26,
          preemph=0.97, srate=16000):
    'Log Mel filter-bank features log(|FFT| @ filters^T + 1) of one signal.'
    fstep, flen_samp = int(srate * frate), int(srate * flen)
    batch = _Batch([signal], flen_samp, fstep)
    filters = create_fbank(nfilters, _fft_len(flen_samp), lowfreq=lowfreq, highfreq=hifreq)
    out, _ = _run(batch, 0, flen_samp, fstep, preemph, np.hamming(flen_samp), filters=filters,
                  apply_log=True, log_offset=1.)
    return out.cpu().numpy()


def extract(signals, conf=None, as_numpy=True):
    '''The `beer features extract` pipeline for a list of signals: magnitude
    spectrum -> filter bank -> log(1e-6 + .) -> cosine transform, liftering ->
    energy -> deltas -> utterance mean normalisation.  Returns one [T_u, dim]
    float64 array per utterance (device tensors with `as_numpy=False`).'''
    cfg = dict(FEACONF)
    for key, val in (conf or {}).items():
        if key not in cfg:
            raise KeyError(f'Unknown setting "{key}"')
        cfg[key] = val
    signals = list(signals)
    if not signals:
        return []
    srate = cfg['srate']
    fstep, flen = int(srate * cfg['framerate']), int(srate * cfg['window_len'])
    batch = _Batch(signals, flen, fstep)
    fft_len = _fft_len(flen)
    filters = None
    if cfg['apply_fbank']:
        filters = create_fbank(cfg['nfilters'], fft_len, lowfreq=cfg['cutoff_lfreq'],
                               highfreq=cfg['cutoff_hfreq'])
    norm = math.sqrt(2. / cfg['nfilters'])
    dct = lift = None
    if cfg['apply_dct']:
        nf = cfg['nfilters'] if cfg['apply_fbank'] else fft_len // 2
        dct = dct_bases(nf, cfg['n_dct_coeff'])
        lift = lifter(cfg['n_dct_coeff'], cfg['lifter_coeff'])
    orders = cfg['delta_order'] if cfg['apply_deltas'] else 0
    buf, base = _run(batch, 1, flen, fstep, cfg['preemph'], np.hamming(flen), filters=filters,
                     apply_log=True, log_offset=1e-6, dct=dct, lift=lift, norm=norm,
                     add_energy=cfg['add_energy'], extra_cols=orders)
    if orders:
        _deltas(buf, batch.frame_off, batch.nutt, base, [cfg['delta_winlen']] * orders)
    if cfg['utt_mnorm']:
        _hip.call('beer_features_cmn', batch.nutt, _hip.ptr(batch.frame_off), buf.shape[1],
                  buf.shape[1], _hip.ptr(buf))
    parts = torch.split(buf, batch.nframes)
    return [p.cpu().numpy() for p in parts] if as_numpy else list(parts)
