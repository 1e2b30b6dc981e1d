"""HIP-event timers of C-ABI calls and phases, and look-ups into the committed profiles."""

import json
import os

import torch

from beer_amd import _hip

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class KernelTimer:
    'HIP-event timing of chosen C-ABI calls on the launching (current) stream.'

    def __init__(self, names):
        self.names, self.events = set(names), {n: [] for n in names}
        self._orig = _hip.call

    def __enter__(self):
        def timed(name, *args):
            if name not in self.names:
                return self._orig(name, *args)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            self._orig(name, *args)
            b.record()
            self.events[name].append((a, b))
        _hip.call = timed
        return self

    def __exit__(self, *exc):
        _hip.call = self._orig

    def mean_ms(self, name):
        ev = self.events[name]
        return sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev)), len(ev)


class PhaseTimer:
    'HIP-event timing of named phases of a step (all-reduce, M-step) on the current stream.'

    def __init__(self):
        self.spans = {}

    def span(self, name):
        timer = self

        class _Span:
            def __enter__(self):
                self.a = torch.cuda.Event(enable_timing=True)
                self.b = torch.cuda.Event(enable_timing=True)
                self.a.record()

            def __exit__(self, *exc):
                self.b.record()
                timer.spans.setdefault(name, []).append((self.a, self.b))
        return _Span()

    def mean_ms(self, name):
        ev = self.spans.get(name, [])
        return sum(a.elapsed_time(b) for a, b in ev) / max(1, len(ev))

    def clear(self):
        self.spans = {}


def pmc_traffic(kernel_key):
    '''HBM bytes per launch of a kernel from the committed PMC passes
    (profiles/r*_pmc.json: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
    passes, full-size launches of this same command).  Counters cannot be read
    from inside the timed run, so this is the last profiled value; None if absent.'''
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc.json')), reverse=True):
        try:
            k = json.load(open(path))['kernels'][kernel_key]
            # FETCH_SIZE under-reports wide (16 B / lane) coalesced reads by 2x on gfx950
            # (MI355X_MICROARCH.md): kernels that stream with 16-byte loads are corrected
            read = k['hbm_read_bytes_raw'] * (2. if k.get('wide_loads', kernel_key.startswith('acc'))
                                              else 1.)
            if not read:
                continue
            return read + k['hbm_write_bytes']
        except Exception:
            continue
    return None


def kernel_times_entry(kernel_key):
    """The committed profile of a kernel's launches (profiles/r*_kernel_times.json, written by
    tools/trace_stats.py from `rocprofv3 --kernel-trace` of the same command at the bench's own
    step counts, warm-up launches dropped), newest round first; ({}, None) if absent."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_kernel_times.json')), reverse=True):
        try:
            return json.load(open(path))['kernels'][kernel_key], os.path.relpath(path, ROOT)
        except Exception:
            continue
    return {}, None


def profiled(kernel_key, roof, per_launch_scale=1.):
    """What the committed profiles say about the roofline's kernel: `traffic` (HBM bytes per
    launch from the PMC passes, `traffic_source`), and the same fraction priced on the PROFILED
    average launch time (`frac_profiled`, `profiled_avg_launch_ms`, `profiled_clock_ghz`): the
    line's own `frac` uses the HIP events of this run."""
    out = {'traffic': None, 'traffic_source': None}
    if not kernel_key:
        return out
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc.json')), reverse=True):
        try:
            json.load(open(path))['kernels'][kernel_key]
        except Exception:
            continue
        t = pmc_traffic(kernel_key)
        if t:
            out['traffic'], out['traffic_source'] = t * per_launch_scale, os.path.relpath(path, ROOT)
        out['profiled_clock_ghz'] = pmc_entry(kernel_key).get('clock_ghz')
        break
    kt, src = kernel_times_entry(kernel_key)
    if kt.get('avg_ms') and roof.get('avg_launch_ms') and roof.get('frac') is not None:
        out['profiled_avg_launch_ms'] = kt['avg_ms'] * per_launch_scale
        out['frac_profiled'] = roof['frac'] * roof['avg_launch_ms'] / (kt['avg_ms'] * per_launch_scale)
        out['profiled_source'] = src
    return out


def pmc_entry(kernel_key):
    'The committed PMC summary of a kernel (profiles/r*_pmc.json), newest round first; {} if absent.'
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc.json')), reverse=True):
        try:
            return json.load(open(path))['kernels'][kernel_key]
        except Exception:
            continue
    return {}


class ClockProbe:
    """The shader clock the kernels of a step really run at (beer_clock_probe: one sleeping wave
    on a side stream reading s_memtime against the 100 MHz s_memrealtime while `work()` keeps
    the main stream busy) -- recorded in the line so that a `frac` measured on one box can be
    compared with `frac_profiled` from another."""

    REF_MHZ = 100.

    def __init__(self, device):
        self.side = torch.cuda.Stream(device=device)
        self.ticks = torch.zeros(2, dtype=torch.int64, device=device)

    def _probe(self, sleeps):
        with torch.cuda.stream(self.side):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            _hip.call('beer_clock_probe', _hip.ptr(self.ticks), sleeps)
            b.record()
        return a, b

    def _read(self, ev):
        torch.cuda.synchronize()
        t, r = (int(v) for v in self.ticks.cpu())
        ms = ev[0].elapsed_time(ev[1])
        return {'mhz': self.REF_MHZ * t / max(1, r),
                # (the reference clock against the HIP events of the launch: ~100 if it is what
                #  the formula assumes)
                'ref_mhz_by_events': r / max(1e-9, ms * 1e3), 'window_ms': r / (self.REF_MHZ * 1e3)}

    def measure(self, work, work_ms):
        """Idle clock, then the clock over a window inside `work()` (which enqueues about
        `work_ms` of kernels on the current stream and does not synchronise)."""
        out = {'idle': self._read(self._probe(100))}
        # the window: the middle ~60 % of two back-to-back works
        per_sleep_ms = out['idle']['window_ms'] / 100.
        sleeps = max(10, int(.6 * work_ms / max(1e-6, per_sleep_ms)))
        work()
        ev = self._probe(sleeps)
        work()
        work()
        out['under_load'] = self._read(ev)
        return out
