"""The lines bench.py prints.

The driver parses the LAST line of stdout, so that line is compact (< 4 KB, `compact()`): the
contract's keys, the headline's `roofline` and `cpu_baseline` as numbers (no prose), and a
`summary` of every other configuration.  Everything else -- per-kernel tables, the other
configurations' own rooflines and baselines, the prose notes -- is the DETAIL object: written to
`bench_detail.json` (and `gpurun_out/bench_detail.json` when that directory exists) and printed
on stderr, one JSON line per sub-object, before the compact line.  Pure Python: importable (and
tested) without torch or a GPU.
"""

import json
import os
import sys

LIMIT = 4096

# the imported reference itself on config 1, seconds per iteration
REFERENCE_CONFIG1_S_PER_ITERATION = 0.160

# (key of the line -> keys kept of that sub-object; None = the value as it is)
COMPACT_KEYS = (
    'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
    'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'ranks', 'backend', 'frames_per_rank',
    'all_reduce_ms', 'm_step_ms', 'elbo_rel_err_vs_cpu_fp64', 'stats_rel_err_vs_cpu_fp64',
    'parity_vs_cpu_fp64', 'f32_mode', 'count_conservation_rel_err', 'roofline', 'cpu_baseline',
    'clock', 'detail', 'summary')
REQUIRED_KEYS = (
    'metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
    'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'ranks', 'backend', 'all_reduce_ms',
    'm_step_ms', 'roofline', 'summary')
ROOFLINE_KEYS = ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'avg_launch_ms',
                 'frac_profiled', 'profiled_avg_launch_ms', 'profiled_clock_ghz', 'traffic_source',
                 'frac_of_bf16_mfma_peak', 'algorithmic_per_launch', 'executed_over_algorithmic')
CPU_BASELINE_KEYS = ('value', 'unit', 'cores', 'kind', 'sample', 'host_threads', 'host_cores',
                     'host_sockets')
CONFIG_KEYS = ('workload', 'parallelism', 'frames_per_gpu', 'components', 'dim', 'frames_total',
               'frames_rank0', 'utterances')


def _num(v, digits=6):
    'Floats at `digits` significant digits (the detail keeps them in full); the rest unchanged.'
    if isinstance(v, bool) or not isinstance(v, float):
        return v
    if v != v or v in (float('inf'), -float('inf')):
        return None
    return float(f'{v:.{digits}g}')


def _numbers(obj, digits=4):
    'A nested dict reduced to its numeric leaves, rounded (parity tables: numbers only).'
    if isinstance(obj, dict):
        out = {k: _numbers(v, digits) for k, v in obj.items()}
        return {k: v for k, v in out.items() if v is not None and v != {}}
    if isinstance(obj, (int, float)) and not isinstance(obj, bool):
        return _num(float(obj), digits) if isinstance(obj, float) else obj
    return None


def _clip(text, n):
    text = ' '.join(str(text).split())
    return text if len(text) <= n else text[:n - 3] + '...'


def summary(out):
    'The other configurations of the default line, as numbers.'
    s = {'config2_frames_per_s': round(out['value']), 'config2_ms_per_step': round(out['ms_per_step'], 3)}
    roof = out.get('roofline') or {}
    if roof.get('frac') is not None:
        s['config2_roofline_frac'] = round(roof['frac'], 4)
    if 'captured' in out:
        s['config2_captured_frames_per_s'] = round(out['captured']['value'])
        s['config2_captured_ms_per_step'] = round(out['captured']['ms_per_step'], 3)
    for key in ('config3', 'config3_full', 'config3_shard'):
        if key in out:
            s[key + '_frames_per_s'] = round(out[key]['value'])
            s[key + '_ms_per_step'] = round(out[key]['ms_per_step'], 3)
            frac = (out[key].get('roofline') or {}).get('frac')
            if frac is not None and key != 'config3_shard':
                s[key + '_roofline_frac'] = round(frac, 4)
            cpu = (out[key].get('cpu_baseline') or {}).get('value')
            if cpu:
                s[key + '_cpu_frames_per_s'] = round(cpu)
    if 'config3_shard' in out and 'projected_8gpu' in out['config3_shard']:
        s['config3_projected_8gpu_speedup'] = round(out['config3_shard']['projected_8gpu']['speedup'], 2)
    c4 = out.get('config4')
    if c4:
        for cov in ('diagonal', 'full'):
            if cov in c4:
                s[f'config4_prior_{cov}_frames_per_s'] = round(c4[cov]['prior_hot_path']['value'])
                s[f'config4_step_{cov}_frames_per_s'] = round(c4[cov]['vae_step']['value'])
                frac = (c4[cov].get('roofline') or {}).get('frac')
                if frac is not None:
                    s[f'config4_prior_{cov}_roofline_frac'] = round(frac, 4)
    c1 = out.get('config1')
    if c1:
        s['config1_us_per_iteration'] = {k: round(c1[k]['us_per_iteration'], 1)
                                         for k in ('eager', 'default', 'captured') if k in c1}
        cpu = c1.get('cpu_baseline') or {}
        if cpu.get('us_per_iteration'):
            s['config1_us_per_iteration']['cpu_port_1_thread'] = round(cpu['us_per_iteration'], 1)
        # the honest neighbour: the imported reference itself
        s['config1_us_per_iteration']['reference_itself'] = REFERENCE_CONFIG1_S_PER_ITERATION * 1e6
    c5 = out.get('config5')
    if c5:
        s['config5_frames_per_s'] = round(c5['value'])
        s['config5_wall_s'] = round(c5['wall_s'], 3)
        if 'training_frames_per_s' in c5:
            s['config5_training_frames_per_s'] = round(c5['training_frames_per_s'])
        if (c5.get('captured_epochs') or {}).get('epoch_ms'):
            s['config5_captured_epoch_ms'] = round(c5['captured_epochs']['epoch_ms'], 3)
        cpu = (c5.get('cpu_baseline') or {}).get('value')
        if cpu:
            s['config5_cpu_frames_per_s'] = round(cpu)
    return s


def compact(out, detail_path=None):
    """The final line of a run as a dict: the contract's keys of `out` (a config-2 or config-3
    line), sub-objects cut to their numeric keys."""
    line = {}
    for key in COMPACT_KEYS:
        if key == 'detail' and detail_path:
            line['detail'] = detail_path
            continue
        if key not in out:
            continue
        v = out[key]
        if key == 'config':
            v = {k: (_clip(v[k], 220) if k == 'workload' else v[k]) for k in CONFIG_KEYS if k in v}
        elif key == 'roofline':
            v = {k: (_num(v[k]) if isinstance(v[k], float) else v[k]) for k in ROOFLINE_KEYS
                 if v.get(k) is not None or k == 'traffic'}
        elif key == 'cpu_baseline':
            v = {k: (_clip(v[k], 150) if k == 'sample' else _num(v[k])) for k in CPU_BASELINE_KEYS
                 if k in v}
        elif key == 'parity_vs_cpu_fp64':
            v = _numbers(v)
        elif key in ('frames_per_rank', 'clock'):
            v = _numbers(v, 6)
        elif key in ('value', 'ms_per_step'):
            pass
        else:
            v = _num(v)
        line[key] = v
    if 'summary' not in line:
        line['summary'] = summary(out)
    return line


def compact_line(out, detail_path=None):
    'The compact line as text; refuses to return more than LIMIT bytes.'
    line = compact(out, detail_path)
    text = json.dumps(line, separators=(', ', ': '))
    if len(text) >= LIMIT:                     # drop the optional pieces, largest first
        for key in ('parity_vs_cpu_fp64', 'clock', 'frames_per_rank'):
            line.pop(key, None)
            text = json.dumps(line, separators=(', ', ': '))
            if len(text) < LIMIT:
                break
    if len(text) >= LIMIT:
        raise ValueError(f'the final bench line is {len(text)} bytes (limit {LIMIT})')
    return text


def emit(out, root, stream_detail=sys.stderr, stream_line=sys.stdout):
    """Write the detail (file + one stderr line per sub-object), then print the compact line --
    the only thing on stdout."""
    paths = [os.path.join(root, 'bench_detail.json')]
    if os.path.isdir(os.path.join(root, 'gpurun_out')):
        paths.append(os.path.join(root, 'gpurun_out', 'bench_detail.json'))
    written = None
    for path in paths:
        try:
            with open(path, 'w') as fh:
                json.dump(out, fh)
            written = written or os.path.relpath(path, root)
        except OSError:
            pass
    head = {k: v for k, v in out.items() if not (isinstance(v, dict) and k.startswith('config')
                                                  and k != 'config')}
    print('BENCH_DETAIL headline ' + json.dumps(head), file=stream_detail, flush=True)
    for k, v in out.items():
        if isinstance(v, dict) and k.startswith('config') and k != 'config':
            print(f'BENCH_DETAIL {k} ' + json.dumps(v), file=stream_detail, flush=True)
    print(compact_line(out, written), file=stream_line, flush=True)
