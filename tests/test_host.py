"""CPU-only checks: the C-ABI library builds, loads and exports every symbol
include/beer_hip.h declares; host-side logic (graph compilation, mean-field
grouping, optimizer round-robin, ELBO bookkeeping); and the product path fails
loudly -- never falls back to the CPU -- when there is no GPU."""

import ctypes
import os
import pickle
import re

import numpy as np
import pytest
import torch

from helpers import ROOT, assert_close, load_golden

import beer_amd as beer
from beer_amd import _hip, build as beer_build

HAS_GPU = torch.cuda.is_available()


def _declared():
    text = open(os.path.join(ROOT, 'include', 'beer_hip.h')).read()
    return sorted(set(re.findall(r'^(?:int|size_t) (beer_\w+)\(', text, flags=re.M)))


def test_library_exports_every_declared_symbol():
    beer_build.build(verbose=False)
    lib = ctypes.CDLL(_hip.LIB_PATH)
    names = _declared()
    assert len(names) >= 35
    for name in names:
        assert hasattr(lib, name), f'{name} is declared but not exported'
    assert sorted(list(_hip.SIGNATURES) + list(_hip.SIZE_QUERIES) +
                  list(_hip.HOST_SIGNATURES)) == names, \
        'ctypes table and header disagree'
    assert lib.beer_hip_version() >= 100


def test_tuning_options_are_clamped():
    '''beer_hip_set_option refuses values outside an option's range (the chain length is a
    divisor; round 3 read it with atoi from the environment) and unknown options.'''
    lib = _hip.lib()
    code = _hip.OPTIONS['ax_max_frames'][0]
    assert lib.beer_hip_get_option(code) == 4096
    assert lib.beer_hip_set_option(code, 0) == _hip.EINVAL
    assert lib.beer_hip_set_option(code, -5) == _hip.EINVAL
    assert lib.beer_hip_set_option(99, 1) == _hip.EINVAL
    assert lib.beer_hip_get_option(99) == _hip.EINVAL
    assert lib.beer_hip_get_option(code) == 4096
    assert _hip.set_option('ax_max_frames', 2048) == 4096
    assert _hip.get_option('ax_max_frames') == 2048
    _hip.set_option('ax_max_frames', 4096)
    with pytest.raises(ValueError):
        _hip.set_option('accf_rounds', 0)


def test_every_option_starts_at_its_documented_default():
    """Each BEER_OPT_* of include/beer_hip.h documents "Default N": a FRESH load of the library
    (its own process: no test has touched the options, no BEER_* preset in the environment)
    must report exactly that for every option, and `_hip.OPTIONS` must name every one of them
    (round 5 shipped BEER_OPT_K1_LDS documented 1, actual 0)."""
    import re
    import subprocess
    import sys
    text = open(os.path.join(ROOT, 'include', 'beer_hip.h')).read()
    count = int(re.search(r'#define\s+BEER_OPT_COUNT\s+(\d+)', text).group(1))
    documented = {}
    for m in re.finditer(r'#define\s+(BEER_OPT_\w+)\s+(\d+)\s*/\*(.*?)\*/', text, re.S):
        if m.group(1) == 'BEER_OPT_COUNT':
            continue
        d = re.search(r'Default\s+(\d+)', m.group(3))
        assert d, f'{m.group(1)}: no "Default N" in its comment'
        documented[int(m.group(2))] = int(d.group(1))
    assert sorted(documented) == list(range(count))
    assert sorted(code for code, _ in _hip.OPTIONS.values()) == list(range(count))
    env = {k: v for k, v in os.environ.items() if not k.startswith('BEER_')}
    out = subprocess.run(
        [sys.executable, '-c',
         'import ctypes, sys; l = ctypes.CDLL(sys.argv[1]); '
         f'print([l.beer_hip_get_option(i) for i in range({count})])', _hip.LIB_PATH],
        env=env, check=True, capture_output=True, text=True).stdout
    assert eval(out) == [documented[i] for i in range(count)]
    # and through the loader the package uses (the environment presets only when set)
    if not any(env_name in os.environ for _, env_name in _hip.OPTIONS.values()):
        for name, (code, _) in _hip.OPTIONS.items():
            if name != 'ax_max_frames':          # (restored by the test above)
                assert _hip.get_option(name) == documented[code], name


def test_struct_layouts_match_header():
    # 4 int32 + 14 pointers; 6 int32 + 6 pointers (LP64)
    assert ctypes.sizeof(_hip.Graph) == 16 + 15 * 8
    assert ctypes.sizeof(_hip.GraphLowDeg) == 8 + 14 * 8
    assert ctypes.sizeof(_hip.Batch) == 24 + 6 * 8 + 16 + 8          # (+ `order`, round 5)
    assert ctypes.sizeof(_hip.FeaConf) == 8 * 4 + 3 * 8 + 6 * 8


@pytest.mark.skipif(HAS_GPU, reason='checks behaviour on a GPU-less host')
def test_product_path_fails_loudly_without_gpu():
    X = torch.randn(20, 3)
    ns = beer.NormalSet.create(X.mean(0), X.var(0), size=4, cov_type='diagonal')
    model = beer.Mixture.create(ns)
    with pytest.raises(_hip.HipUnavailable):
        beer.evidence_lower_bound(model, X)
    with pytest.raises(_hip.HipUnavailable):
        ns.means_precisions.posterior.expected_sufficient_statistics()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'beer_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in text.replace('the oracle', ''), f


def _notebook_graph():
    graph = beer.graph.Graph()
    s0, s4 = graph.add_state(), graph.add_state()
    graph.start_state, graph.end_state = s0, s4
    s1, s2, s3 = (graph.add_state(pdf_id=i) for i in range(3))
    for a, b in [(s0, s1), (s1, s1), (s1, s2), (s2, s2), (s2, s3), (s3, s3), (s3, s1),
                 (s1, s4), (s2, s4), (s3, s4)]:
        graph.add_arc(a, b)
    graph.normalize()
    return graph.compile()


def _unit(topology, start_pdf_id):
    'Left-to-right unit HMM as `beer hmm mkphones` builds it.'
    ids = sorted({a for a, _, _ in topology} | {b for _, b, _ in topology})
    graph = beer.graph.Graph()
    count = 0
    for sid in range(len(ids)):
        if sid in (ids[0], ids[-1]):
            graph.add_state()
        else:
            graph.add_state(pdf_id=start_pdf_id + count)
            count += 1
    graph.start_state, graph.end_state = ids[0], ids[-1]
    for arc in topology:
        graph.add_arc(*arc)
    return graph, start_pdf_id + count


SIL = [(0, 1, 1.), (1, 1, .5), (1, 2, .5), (2, 2, .5), (2, 1, .25), (2, 3, .25)]
SPEECH = [(0, 1, 1.), (1, 1, .75), (1, 2, .25), (2, 2, .75), (2, 3, .25), (3, 3, .75),
          (3, 4, .25)]


def _units():
    units, pdf = {}, 0
    units['sil'], pdf = _unit(SIL, pdf)
    for name in 'abcd':
        units[name], pdf = _unit(SPEECH, pdf)
    return units


def test_graph_compile_matches_reference():
    g = load_golden('g12_graph_compile')
    nb = _notebook_graph()
    assert_close(nb.init_log_probs.numpy(), g['notebook.init'], 1e-7)
    assert_close(nb.final_log_probs.numpy(), g['notebook.final'], 1e-7)
    assert_close(nb.trans_log_probs.exp().numpy(), np.exp(g['notebook.trans']), 1e-7)
    assert nb.pdf_id_mapping == g['notebook.pdf_id_mapping'].tolist()

    units = _units()
    # alignment graph of `sil a c a sil` (mkaligraph.create_graph_from_seq)
    graph = beer.graph.Graph()
    graph.start_state = graph.add_state()
    last, phone_states = graph.start_state, []
    seq = ['sil', 'a', 'c', 'a', 'sil']
    for phone in seq:
        state = graph.add_state()
        phone_states.append(state)
        graph.add_arc(last, state)
        last = state
    graph.end_state = graph.add_state()
    graph.add_arc(last, graph.end_state)
    for state, phone in zip(phone_states, seq):
        graph.replace_state(state, units[phone])
    graph.normalize()
    ali = graph.compile()
    assert ali.pdf_id_mapping == g['ali.pdf_id_mapping'].tolist()
    assert_close(ali.init_log_probs.exp().numpy(), np.exp(g['ali.init']), 1e-7)
    assert_close(ali.final_log_probs.exp().numpy(), np.exp(g['ali.final']), 1e-7)
    assert_close(ali.trans_log_probs.exp().numpy(), np.exp(g['ali.trans']), 1e-6)

    # phone-loop decoding graph (mkphoneloopgraph + mkdecodegraph)
    graph = beer.graph.Graph()
    graph.start_state, graph.end_state = graph.add_state(), graph.add_state()
    pivot = graph.add_state()
    u2s = {name: graph.add_state() for name in units}
    graph.add_arc(graph.start_state, u2s['sil'])
    graph.add_arc(u2s['sil'], graph.end_state)
    for name in units:
        graph.add_arc(pivot, u2s[name])
        graph.add_arc(u2s[name], pivot)
    graph.normalize()
    for name, hmm in units.items():
        graph.replace_state(u2s[name], hmm)
    graph.normalize()
    loop = graph.compile()
    assert loop.pdf_id_mapping == g['ploop.pdf_id_mapping'].tolist()
    assert_close(loop.init_log_probs.exp().numpy(), np.exp(g['ploop.init']), 1e-7)
    assert_close(loop.final_log_probs.exp().numpy(), np.exp(g['ploop.final']), 1e-7)
    # (the golden's loop-back entries were rewritten by PhoneLoop's callback;
    #  compare everything else)
    ref = np.exp(g['ploop.trans'])
    got = loop.trans_log_probs.exp().numpy()
    mask = np.ones_like(ref, dtype=bool)
    for e in g['end_idxs']:
        mask[e, g['start_idxs']] = False
    assert_close(got[mask], ref[mask], 1e-6)


def test_mean_field_grouping_and_round_robin():
    X = torch.randn(50, 3)
    ns = beer.NormalSet.create(X.mean(0), X.var(0), size=4, cov_type='diagonal')
    model = beer.Mixture.create(ns)
    groups = model.mean_field_factorization()
    assert len(groups) == 1 and len(groups[0]) == 2                 # Q7: one merged group
    assert list(model.bayesian_parameters()) == groups[0]
    assert [len(g) for g in model.conjugate_bayesian_parameters(keepgroups=True)] == [2]

    class Fake:
        def __init__(self):
            self.updates, self.zeroed = 0, 0

        def natural_grad_update(self, lrate):
            self.updates += 1

        def zero_stats(self):
            self.zeroed += 1

    a, b, c = Fake(), Fake(), Fake()
    optim = beer.VBConjugateOptimizer([[a, b], [c]], lrate=.5)
    optim.init_step()
    assert (a.zeroed, b.zeroed, c.zeroed) == (1, 1, 1)
    optim.step()
    assert (a.updates, b.updates, c.updates) == (1, 1, 0)
    optim.step()
    assert (a.updates, b.updates, c.updates) == (1, 1, 1)
    state = optim.state_dict()
    assert state == {'lrate': .5, 'update_count': 2}
    other = beer.VBConjugateOptimizer([[a, b], [c]])
    other.load_state_dict(state)
    other.step()
    assert a.updates == 2 and c.updates == 1


def test_elbo_object_bookkeeping():
    X = torch.randn(50, 3)
    ns = beer.NormalSet.create(X.mean(0), X.var(0), size=4, cov_type='diagonal')
    p = ns.means_precisions
    q = beer.Mixture.create(ns).categorical.weights
    E = beer.EvidenceLowerBoundInstance
    e1 = E(torch.tensor(-3.), {p: torch.ones(4, 8)}, [p], 10, 100)
    e2 = E(torch.tensor(-4.), {p: torch.ones(4, 8), q: torch.ones(4)}, [p, q], 30, 100)
    s = beer.evidence_lower_bound(datasize=100) + e1 + e2
    assert float(s) == -7. and s._minibatchsize == 40
    assert torch.equal(s._acc_stats[p], 2 * torch.ones(4, 8))
    s.backward()                                                     # Q2: scale N / sum T_b
    assert torch.equal(p.stats, 5. * torch.ones(4, 8))
    assert torch.equal(q.stats, 2.5 * torch.ones(4))
    with pytest.raises(ValueError):
        e1 + E(torch.tensor(0.), {}, [], 1, 99)
    with pytest.raises(ValueError):
        beer.evidence_lower_bound(None, X)
    # parameters are keyed by a uuid that survives pickling (elbo.sync)
    p2 = pickle.loads(pickle.dumps(p))
    assert p2 == p and hash(p2) == hash(p)


def test_default_priors_follow_the_reference_recipe():
    torch.manual_seed(0)
    mean, var = torch.tensor([1., -2., .5]), torch.tensor([2., .5, 1.])
    ns = beer.NormalSet.create(mean, var, size=5, prior_strength=2., noise_std=0.,
                               cov_type='full')
    pr = ns.means_precisions.prior.params
    assert pr.scale.shape == (5, 1) and float(pr.scale[0]) == 2.
    assert float(pr.dof[0]) == 2. + 3 - 1
    assert torch.allclose(pr.scale_matrix[0], torch.diag(1 / var) / 4.)
    assert torch.equal(ns.means_precisions.posterior.params.mean, pr.mean)
    nd = beer.NormalSet.create(mean, var, size=5, prior_strength=2., cov_type='diagonal')
    assert torch.allclose(nd.means_precisions.prior.params.rates[0], 2. * var)
    ni = beer.NormalSet.create(mean, var, size=5, prior_strength=2., cov_type='isotropic')
    assert float(ni.means_precisions.prior.params.rate[0]) == 4.
    assert ni.means_precisions.stats.shape == (5, 6)
    assert nd.means_precisions.stats.shape == (5, 8)
    assert ns.means_precisions.stats.shape == (5, 14)
    m = beer.Mixture.create(ns, prior_strength=3.)
    assert torch.allclose(m.categorical.weights.prior.params.concentrations,
                          torch.full((5,), 3. / 5))


def test_native_compile_equals_python_oracle_and_batch_builder():
    '''beer_graph_compile / beer_aligraphs_compile against the plain-Python
    restatement (oracle/graph_oracle.py) on random transcriptions, and the
    one-call corpus builder against per-utterance compilation.'''
    sys_path_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import sys
    sys.path.insert(0, sys_path_root)
    from oracle import graph_oracle as go
    units = _units()
    rng = np.random.RandomState(0)
    names = list(units)
    seqs = [[names[i] for i in rng.randint(0, len(names), n)] for n in (1, 2, 5, 17, 40)]
    gset = beer.graph.compile_alignments(seqs, units)
    assert len(gset) == len(seqs)
    for seq, sparse in zip(seqs, gset):
        graph = go.alignment_graph(seq, units, beer.graph.Graph)
        init, final, trans, pdf = go.compile_graph(graph)
        one = graph.compile()                                       # native, one graph
        assert one.pdf_id_mapping == pdf == [int(i) for i in sparse.pdf_id_mapping]
        for got in (one, sparse.to_dense()):
            assert_close(got.init_log_probs.exp().numpy(), init, 1e-7)
            assert_close(got.final_log_probs.exp().numpy(), final, 1e-7)
            assert_close(got.trans_log_probs.exp().numpy(), trans, 1e-6)
        assert sparse.n_states == len(pdf) == 3 * len(seq) - sum(u == 'sil' for u in seq)
    # the phone loop: a non-emitting pivot reached from every unit
    loop = beer.graph.Graph()
    loop.start_state, loop.end_state = loop.add_state(), loop.add_state()
    pivot = loop.add_state()
    u2s = {name: loop.add_state() for name in units}
    loop.add_arc(loop.start_state, pivot)
    loop.add_arc(pivot, loop.end_state)
    for name in units:
        loop.add_arc(pivot, u2s[name])
        loop.add_arc(u2s[name], pivot)
    loop.normalize()
    for name, hmm in units.items():
        loop.replace_state(u2s[name], hmm)
    loop.normalize()
    init, final, trans, pdf = go.compile_graph(loop)
    got = loop.compile()
    assert got.pdf_id_mapping == pdf
    assert_close(got.init_log_probs.exp().numpy(), init, 1e-7)
    assert_close(got.final_log_probs.exp().numpy(), final, 1e-7)
    assert_close(got.trans_log_probs.exp().numpy(), trans, 1e-6)
    with pytest.raises(_hip.HipError):
        beer.graph.compile_alignments([[]], units)                  # empty transcription


def test_f32_mode_is_a_per_call_flag_not_library_state():
    '''The library exports no mode setter: the arithmetic of float32 products is
    the BEER_EXACT bit of each call's `dtype`; the host-side preference lives in
    beer_amd._hip (initial value from BEER_F32_MODE).'''
    lib = ctypes.CDLL(_hip.LIB_PATH)
    assert not hasattr(lib, 'beer_hip_set_f32_mode')
    assert _hip.dtype_code(torch.float32) == _hip.F32
    assert _hip.dtype_code(torch.float32, exact=True) == _hip.F32 | _hip.EXACT
    assert _hip.dtype_code(torch.float64, exact=True) == _hip.F64
    old = _hip.get_f32_mode()
    for mode in ('exact', 'bf16x3'):
        _hip.set_f32_mode(mode)
        assert _hip.get_f32_mode() == mode
    with pytest.raises(ValueError):
        _hip.set_f32_mode('bf16')
    _hip.set_f32_mode(old)
    with _hip.exact_f32():
        assert _hip.get_f32_mode() == 'exact'
    assert _hip.get_f32_mode() == old


def test_fast_f32_path_needs_no_look_at_the_data():
    '''The bf16x3 arithmetic holds every float32 operand exactly (three bf16 pieces,
    fp32's exponent range): whether a tensor takes it depends on dtype, size and the
    host-side mode only -- no range check on the device, no synchronisation, nothing
    to memoise (round 2's fp16 split needed all three).'''
    calls = []
    orig = _hip.call
    _hip.call = lambda name, *args: calls.append(name)
    try:
        a = torch.zeros(_hip.FAST_MIN_FRAMES + 8, 4)
        assert _hip.f32_fast_ok(a) and _hip.f32_fast_ok(a[8:])
        assert not _hip.f32_fast_ok(a[:100])                         # small: exact kernels
        assert not _hip.f32_fast_ok(a.double())
        assert not _hip.f32_fast_ok(torch.zeros(_hip.FAST_MIN_FRAMES, _hip.MAX_DIM_FAST + 1))
        huge = a.clone()
        huge[0, 0] = 3e38                                            # any float32 value will do
        huge[1, 1] = 1e-30
        assert _hip.f32_fast_ok(huge)
        with _hip.exact_f32():
            assert not _hip.f32_fast_ok(a)
        assert _hip.f32_fast_ok(a)
    finally:
        _hip.call = orig
    assert calls == []
    lib = ctypes.CDLL(_hip.LIB_PATH)
    assert not hasattr(lib, 'beer_f32_split_hazard') and not hasattr(lib, 'beer_frame_scales')


@pytest.mark.parametrize('T', [64, 70, 333])
def test_row_split_weight_gradient_of_the_vae_layers_is_nn_linears(T, monkeypatch):
    '''`beer_amd.nnet.Linear` (the layers of the residual nets and the VAE's heads): above
    `ROW_SPLIT_MIN` rows the weight gradient is a batched product over blocks of rows plus a tail;
    forward, input gradient and bias gradient are nn.Linear's, the weight gradient equals it to
    float64 rounding, state dict and pickle keep nn.Linear's names.'''
    from beer_amd.nnet import linear
    monkeypatch.setattr(linear, 'ROW_SPLIT_MIN', 64)
    monkeypatch.setattr(linear, 'ROW_BLOCK', 16)
    torch.manual_seed(T)
    lin = beer.nnet.Linear(7, 5).double()
    ref = torch.nn.Linear(7, 5).double()
    ref.load_state_dict(lin.state_dict())
    x = torch.randn(T, 7, dtype=torch.float64, requires_grad=True)
    x2 = x.detach().clone().requires_grad_(True)
    y, y2 = lin(x), ref(x2)
    assert y.grad_fn.name().startswith('_RowSplitLinear') and torch.equal(y, y2)
    y.tanh().sum().backward()
    y2.tanh().sum().backward()
    assert torch.equal(x.grad, x2.grad) and torch.equal(lin.bias.grad, ref.bias.grad)
    assert_close(lin.weight.grad.numpy(), ref.weight.grad.numpy(), 1e-13, 'weight gradient')
    with torch.no_grad():
        assert lin(x).grad_fn is None
    net = pickle.loads(pickle.dumps(beer.nnet.ResidualFeedForwardNet(4, 1, 3)))
    assert sorted(net.state_dict()) == ['blocks.0.layer1.bias', 'blocks.0.layer1.weight',
                                        'blocks.0.layer2.bias', 'blocks.0.layer2.weight']


def test_frames_workspace_of_the_diagonal_accumulation_is_bounded():
    '''`beer_accumulate_frames_workspace_bytes`: never below the shape's own query; for the
    float32 diagonal accumulation over frames it holds per-chain partial sums (grows with T) --
    but never more than 512 MiB (beyond, the call flushes with atomics), and not at all for
    BEER_EXACT, float64, full covariance or fewer than 16384 frames.'''
    lib = _hip.lib()
    q, base = lib.beer_accumulate_frames_workspace_bytes, lib.beer_accumulate_workspace_bytes
    F32, F64, FULL, DIAG, ISO = _hip.F32, _hip.F64, 0, 1, 2
    b = base(F32, DIAG, 64, 120, 1)
    small = q(F32, DIAG, 1_000_000, 64, 120, 1)
    assert small > b and small == 489 * 128 * (128 + 1) * 4 + 256            # 32 MB at config 4
    assert q(F32, ISO, 1_000_000, 24, 48, 1) == 489 * 64 * (2 * 32 + 1) * 4 + 256
    assert q(F32, DIAG, 10_000_000, 64, 65536, 1) == base(F32, DIAG, 64, 65536, 1)  # would be 165 GB
    assert q(F32, DIAG, 16_383, 64, 120, 1) == b
    assert q(F32 | _hip.EXACT, DIAG, 1_000_000, 64, 120, 1) == base(F32 | _hip.EXACT, DIAG, 64, 120, 1)
    assert q(F64, DIAG, 1_000_000, 64, 120, 1) == base(F64, DIAG, 64, 120, 1)
    assert q(F32, FULL, 1_000_000, 40, 256, 1) == base(F32, FULL, 40, 256, 1)


def test_cli_model_builders_on_the_recipe_configuration():
    """`beer hmm mkphones / mkphoneloopgraph / mkdecodegraph` as builders (beer_amd/cli/hmm.py) on
    the configuration of recipes/aud/conf/hmm.yml (benchlib/recipe.py): unit graphs, pdf ids running
    through the groups, the loop's edge units, first / last pdf of every unit; and
    `phones_of_path` (decode.py:27-40) against the loop it restates."""
    import sys
    sys.path.insert(0, ROOT)
    from benchlib import recipe
    from beer_amd.cli import hmm as cli
    conf = {g['group_name']: g for g in recipe.hmm_conf()}
    topo = cli.UnitTopology(conf['non-speech-unit']['topology'])
    assert (topo.n_states, topo.n_emitting, len(topo.src)) == (7, 5, 19)
    assert cli.UnitTopology(conf['speech-unit']['topology']).n_emitting == 3
    with pytest.raises(ValueError):
        cli.UnitTopology([{'start_id': 0, 'end_id': 2, 'trans_prob': 1.}])      # ids with a hole
    names = {'non-speech-unit': ['sil'], 'speech-unit': ['a', 'b', 'c']}
    torch.manual_seed(0)
    units, emissions = cli.build_units(conf, names, torch.zeros(6), torch.ones(6))
    assert list(units) == ['sil', 'a', 'b', 'c']
    sets = emissions.modelsets
    assert [(len(m), m.n_comp_per_mixture) for m in sets] == [(5, 10), (9, 4)]
    pdfs = {n: [units[n].state_from_id(s).pdf_id for s in units[n].states()] for n in units}
    assert pdfs['sil'] == [None, 0, 1, 2, 3, 4, None]
    assert pdfs['a'] == [None, 5, 6, 7, None] and pdfs['c'] == [None, 11, 12, 13, None]
    loop = cli.loop_graph(list(units), edge_units=['sil'])
    assert sorted(loop.symbols.values(), key=str) == sorted(['\\<s\\>', '\\</s\\>', '#1', 'sil', 'a', 'b', 'c'], key=str)
    arcs = {(a.start, a.end) for a in loop.arcs()}
    state_of = {u: s for s, u in loop.symbols.items()}
    assert (loop.start_state, state_of['sil']) in arcs and (state_of['sil'], loop.end_state) in arcs
    assert (loop.start_state, state_of['a']) not in arcs
    for u in units:
        assert (state_of['#1'], state_of[u]) in arcs and (state_of[u], state_of['#1']) in arcs
    graph, first, last = cli.decode_graph(loop, units)
    assert first == {'sil': 0, 'a': 5, 'b': 8, 'c': 11} and last == {'sil': 4, 'a': 7, 'b': 10, 'c': 13}
    compiled = graph.compile()
    assert compiled.n_states == 14
    # phones_of_path against the reference's loop, written out
    rng = np.random.RandomState(0)
    starts = list(first.values())
    sym = {v: k for k, v in first.items()}
    for _ in range(20):
        path = [0] + [int(v) for v in rng.randint(0, 14, 40)]
        want_per_frame, want = [sym[path[0]]], [sym[path[0]]]
        prev, cur = path[0], sym[path[0]]
        for p in path[1:]:
            if p != prev and p in starts:
                cur = sym[p]
                want.append(cur)
                want_per_frame.append(cur)
            else:
                want_per_frame.append(cur)
            prev = p
        assert cli.phones_of_path(path, first) == want
        assert cli.phones_of_path(path, first, per_frame=True) == want_per_frame
    with pytest.raises(KeyError):
        cli.phones_of_path([3, 0, 1], first)
