"""The final line of bench.py (benchlib/format.py): compact enough for the driver to parse
(round 5's 25 KB line was not: BENCH_r05.json `parsed: null`) and carrying every key of the
contract.  The canned result is the full round-5 line as the GPU box printed it."""

import io
import json
import os

import pytest

from benchlib import format as fmt

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _canned(name='r05_bench.json'):
    return json.load(open(os.path.join(ROOT, 'profiles', name)))


def test_compact_line_is_small_and_complete():
    out = _canned()
    assert len(json.dumps(out)) > 20000               # (the object that broke the parser)
    out.pop('summary', None)
    out['summary'] = fmt.summary(out)
    text = fmt.compact_line(out, 'bench_detail.json')
    assert len(text) < 4096 and '\n' not in text
    line = json.loads(text)
    for key in fmt.REQUIRED_KEYS + ('cpu_baseline', 'elbo_rel_err_vs_cpu_fp64', 'parity_vs_cpu_fp64'):
        assert key in line, key
    # the numbers are the detail's own
    assert line['value'] == out['value'] and line['ms_per_step'] == out['ms_per_step']
    assert line['metric'] == out['metric'] and line['n_gpus'] == 1
    assert line['config']['workload'].startswith('configs[1]: GMM K=256 full-covariance, D=40')
    roof = line['roofline']
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic'):
        assert key in roof, key
    assert roof['frac'] == pytest.approx(out['roofline']['frac'], rel=1e-5)
    assert 'note' not in roof and all(not isinstance(v, str) or len(v) < 64 for v in roof.values())
    cpu = line['cpu_baseline']
    for key in ('value', 'unit', 'cores', 'kind', 'sample'):
        assert key in cpu, key
    assert len(cpu['sample']) <= 150
    # parity: numbers only
    def leaves(o):
        for v in o.values():
            if isinstance(v, dict):
                yield from leaves(v)
            else:
                yield v
    assert all(isinstance(v, (int, float)) for v in leaves(line['parity_vs_cpu_fp64']))
    # the summary names every other configuration and puts the reference itself beside config 1
    s = line['summary']
    for key in ('config3_frames_per_s', 'config3_full_frames_per_s', 'config3_shard_ms_per_step',
                'config4_prior_full_frames_per_s', 'config5_frames_per_s', 'config1_us_per_iteration'):
        assert key in s, key
    assert s['config1_us_per_iteration']['reference_itself'] == 160000.0


def test_config3_line_is_compact_too():
    out = _canned('r05_bench_config3.json')
    out['summary'] = {'config3_frames_per_s': round(out['value'])}
    text = fmt.compact_line(out)
    assert len(text) < 4096
    line = json.loads(text)
    for key in fmt.REQUIRED_KEYS:
        assert key in line, key
    assert line['scaling'] == 'strong'


def test_emit_puts_one_line_on_stdout_and_the_rest_elsewhere(tmp_path):
    out = _canned()
    out['summary'] = fmt.summary(out)
    so, se = io.StringIO(), io.StringIO()
    fmt.emit(out, str(tmp_path), stream_detail=se, stream_line=so)
    lines = so.getvalue().splitlines()
    assert len(lines) == 1 and len(lines[0]) < 4096
    assert json.loads(lines[0])['detail'] == 'bench_detail.json'
    detail = json.load(open(tmp_path / 'bench_detail.json'))
    assert detail['config3']['kernels'] == out['config3']['kernels']
    tags = [ln.split(' ', 2)[1] for ln in se.getvalue().splitlines()]
    assert tags[0] == 'headline' and 'config3' in tags and 'config5' in tags
    for ln in se.getvalue().splitlines():
        json.loads(ln.split(' ', 2)[2])


def test_an_oversized_line_is_refused():
    out = _canned()
    out['summary'] = {f'k{i}': 'x' * 50 for i in range(100)}
    with pytest.raises(ValueError):
        fmt.compact_line(out)
