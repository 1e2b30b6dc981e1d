"""The `cpu_baseline` legs of bench.py: the reference's CPU op sequences (oracle/torch_port.py,
oracle/*_oracle.py) timed on the host cores on bounded samples, and the fp64 parity check of the
headline's kernels.  Together with tests/ and __graft_entry__.smoke() the only code that imports
oracle/ -- as the checker and the reported baseline, never inside a timed GPU region."""

import os
import time

import numpy as np
import torch

import beer_amd as beer

from .shapes import D, K, LATENT, N_COMP, N_PHONES


def host_cores():
    '''The host the CPU baseline ran on: hardware threads, physical cores and sockets
    (/proc/cpuinfo); `cores` of a cpu_baseline is the number of threads the baseline
    actually used (the best of the thread counts probed), these say out of how many.'''
    out = {'host_threads': os.cpu_count() or 1}
    try:
        phys, sockets = set(), set()
        pid = cid = None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                pid = line.split(':')[1].strip()
                sockets.add(pid)
            elif line.startswith('core id'):
                cid = line.split(':')[1].strip()
            elif not line.strip():
                if pid is not None and cid is not None:
                    phys.add((pid, cid))
                pid = cid = None
        if phys:
            out['host_cores'] = len(phys)
            out['host_sockets'] = len(sockets)
            out['host_cores_per_socket'] = len(phys) // max(1, len(sockets))
    except OSError:
        pass
    return out


def cpu_baseline_gmm(frames_target=1 << 20, chunk=8192, budget_s=15.):
    '''beer's CPU path on the host cores: the reference's own op sequence replayed
    with torch CPU ops (oracle/torch_port.py; numerically identical to the
    reference) on a bounded sample of config 2.'''
    from oracle import torch_port as tp
    g = torch.Generator().manual_seed(3)
    n = 16 * chunk
    means = torch.randn(K, D, generator=g) * 2
    X = means[torch.randint(0, K, (n,), generator=g)] + torch.randn(n, D, generator=g)
    mean, cov = X.mean(0), torch.cov(X.t())
    dof = torch.full((K, 1), float(D))
    prior = (mean.repeat(K, 1), torch.ones(K, 1), (cov.inverse() / D).repeat(K, 1, 1), dof)
    post = (prior[0] + torch.randn(K, D, generator=g) * cov.diag().sqrt(),) + prior[1:]
    w = torch.full((K,), 1. / K)
    # Pick the thread count that serves the reference's op mix best on this
    # host (all hardware threads is usually NOT it: the element-wise passes
    # thrash).  The baseline is then timed at that setting.
    ncpu = os.cpu_count() or 1
    best = (0., torch.get_num_threads())
    probe = {}
    for nt in sorted({min(ncpu, c) for c in (8, 16, 32, 64, ncpu)}):
        torch.set_num_threads(nt)
        tp.gmm_elbo(X[:chunk], post, prior, w, w, n)                   # warm-up
        t = time.perf_counter()
        tp.gmm_elbo(X[chunk:2 * chunk], post, prior, w, w, n)
        rate = chunk / (time.perf_counter() - t)
        probe[str(nt)] = rate
        if rate > best[0]:
            best = (rate, nt)
    torch.set_num_threads(best[1])
    t0 = time.perf_counter()
    done, acc_n, acc_w = 0, 0., 0.
    while done < frames_target and time.perf_counter() - t0 < budget_s:
        lo = done % n
        _, an, aw = tp.gmm_elbo(X[lo:lo + chunk], post, prior, w, w, n)
        acc_n, acc_w = acc_n + an, acc_w + aw
        done += chunk
    tp.gmm_update(post, prior, w, w, acc_n * (n / done), acc_w * (n / done), D)
    dt = time.perf_counter() - t0
    return {'value': done / dt, 'unit': 'frames/s', 'cores': int(torch.get_num_threads()),
            **host_cores(), 'kind': 'port',
            'frames_per_s_by_threads': probe,
            'threads_note': '`cores` = the thread count that served the reference\'s op mix best '
                            '(one 8192-frame utterance per count, `frames_per_s_by_threads`: also '
                            'one whole socket, 64 threads); `value` is timed at that count',
            'sample': f'{done} frames of the config-2 workload (K=256 full-cov, D=40, fp32) in '
                      f'{chunk}-frame utterances + 1 M-step, torch-CPU replay of the '
                      f'reference op sequence, {dt:.1f} s'}


def cpu_baseline_features(signals, conf=None):
    '''cpu_baseline leg of tools/bench_features.py: the numpy oracle of the
    feature front-end on a bounded sample of utterances, one host core.
    Returns (frames per second, list of feature matrices).'''
    from oracle import features_oracle as fo
    t0 = time.perf_counter()
    feats = [fo.extract(sig, conf) for sig in signals]
    dt = time.perf_counter() - t0
    return sum(len(f) for f in feats) / dt, feats


def cpu_baseline_graph_compile(sequences, units, graph_cls):
    '''cpu_baseline leg of tools/bench_hmm.py: alignment graphs of a bounded
    sample of transcriptions with the plain-Python restatement of the
    reference's builder + Graph.compile (oracle/graph_oracle.py), one host
    core.  Returns seconds per utterance.'''
    from oracle import graph_oracle as go
    t0 = time.perf_counter()
    for seq in sequences:
        go.compile_graph(go.alignment_graph(seq, units, graph_cls))
    return (time.perf_counter() - t0) / max(1, len(sequences))


def gmm_parity_check(model, X, n=65536, chunk=8192):
    '''The kernels the timed loop runs (packed hand-over: n >= 16384 frames) against
    the fp64 numpy oracle on the same frames, in BOTH float32 arithmetics: relative error of
    the ELBO and of the accumulated statistics -- the whole array and per block (counts,
    first moments, second moments, each against its own largest entry) plus the mean
    relative bias of the counts (the matrix core's truncating accumulate shows there
    first).  Returns {mode: {...}}.'''
    from oracle import beer_oracle as orc
    p0, p1 = list(model.bayesian_parameters())

    def as64(d):
        return [getattr(d.params, nm).cpu().numpy().astype(np.float64)
                for nm in d._std_params_def]
    post, prior, w_post, w_prior = as64(p0.posterior), as64(p0.prior), as64(p1.posterior)[0], \
        as64(p1.prior)[0]
    Xh = X[:n].cpu().numpy().astype(np.float64)
    per_frame, acc_n, kl = 0., 0., None
    for lo in range(0, n, chunk):
        r = orc.gmm_elbo_step(Xh[lo:lo + chunk], 'full', post, prior, w_post, w_prior)
        per_frame += r['per_frame'].sum()
        acc_n, kl = acc_n + r['acc_normal'], r['kl']
    truth = per_frame - kl

    def rel(a, b):
        return float(np.abs(a - b).max() / np.abs(b).max())
    out = {}
    for mode in ('bf16x3', 'exact'):
        old = beer.get_f32_mode()
        beer.set_f32_mode(mode)
        try:
            elbo = beer.accumulate_elbo(model, (X[:n], [n]), datasize=n)
        finally:
            beer.set_f32_mode(old)
        got = elbo._acc_stats[p0].cpu().numpy().astype(np.float64)
        out[mode] = {'elbo_rel_err': abs(float(elbo) - truth) / abs(truth),
                     'stats_rel_err': rel(got, acc_n),
                     'first_moments_rel_err': rel(got[:, :D], acc_n[:, :D]),
                     'second_moments_rel_err': rel(got[:, D:-2], acc_n[:, D:-2]),
                     'counts_rel_err': rel(got[:, -2:], acc_n[:, -2:]),
                     'counts_mean_rel_bias': float(((got[:, -2] - acc_n[:, -2]) / acc_n[:, -2]).mean())}
    return out


def cpu_baseline_hmm(budget_s=20.):
    '''beer's CPU path for config 3 on the host cores: the reference's op sequence for one
    `evidence_lower_bound(PhoneLoop, utterance)` per utterance -- phi(X), stats @ E[T]^T,
    per-state logsumexp, a Python loop of dense [S, S] logsumexp per frame for forward
    and backward, [T-1, S, S] transition posteriors, joint responsibilities^T @ stats,
    KL of every parameter per utterance -- replayed with torch CPU ops
    (oracle/torch_port.py: hmm_elbo) on a bounded sample of utterances.'''
    from oracle import torch_port as tp
    g = torch.Generator().manual_seed(5)
    S, G = 3 * N_PHONES, N_COMP
    KK = S * G
    prior = (torch.zeros(KK, D), torch.ones(KK, 1), torch.ones(KK, 1), torch.ones(KK, D))
    post = (torch.randn(KK, D, generator=g),) + prior[1:]
    w = torch.ones(S, G)
    trans = torch.full((S, S), -float('inf'))
    for s in range(S):
        trans[s, s] = np.log(.75)
        if s % 3 < 2:
            trans[s, s + 1] = np.log(.25)
        else:
            trans[s, ::3] = np.log(.25 / N_PHONES)
    init = torch.where(torch.arange(S) % 3 == 0, torch.tensor(np.log(1. / N_PHONES)),
                       torch.tensor(-float('inf'))).float()
    fin = torch.where(torch.arange(S) % 3 == 2, torch.tensor(np.log(.25)),
                      torch.tensor(-float('inf'))).float()
    rng = np.random.RandomState(2)
    ncpu = os.cpu_count() or 1
    best = (0., torch.get_num_threads())
    probe = torch.randn(300, D, generator=g)
    for nt in sorted({min(ncpu, c) for c in (4, 8, 16)}):
        torch.set_num_threads(nt)
        t = time.perf_counter()
        tp.hmm_elbo(probe, post, prior, w, w, init, fin, trans, 10_000_000)
        rate = 300 / (time.perf_counter() - t)
        if rate > best[0]:
            best = (rate, nt)
    torch.set_num_threads(best[1])
    t0 = time.perf_counter()
    frames = utts = 0
    acc = 0.
    while time.perf_counter() - t0 < budget_s:
        T = int(rng.randint(200, 401))
        X = torch.randn(T, D, generator=g)
        _, a, _, _ = tp.hmm_elbo(X, post, prior, w, w, init, fin, trans, 10_000_000)
        acc = acc + a
        frames += T
        utts += 1
    dt = time.perf_counter() - t0
    return {'value': frames / dt, 'unit': 'frames/s', 'cores': int(torch.get_num_threads()),
            **host_cores(), 'kind': 'port',
            'sample': f'{utts} utterances ({frames} frames) of the config-3 workload (phone loop '
                      f'40x3 states, 16 diagonal Gaussians per state, D=40, fp32): the per-utterance '
                      f'evidence_lower_bound calls of accumulate.py:39-59 (E-step, forward-backward, '
                      f'statistics, KL), torch-CPU replay of the reference op sequence, {dt:.1f} s; '
                      f'the once-per-iteration update of update.py:41-62 (1920 diagonal posteriors, '
                      f'< 1 ms on the host against {1e7 / (frames / dt):.0f} s of accumulation for '
                      f'the 10 M frames) is not in the sample'}


def cpu_baseline_vae_prior(cov, budget_s=8.):
    """The prior hot path of config 4 on the host: per utterance phi(z), stats @ E[T]^T, the Python
    forward-backward loop over the dense 120 x 120 matrix, autograd back to the samples,
    gamma^T @ stats (oracle/torch_port.py: vae_hmm_prior_path, the reference's op sequence,
    vae.py:63-86 / hmm.py:73-100) on a bounded sample of utterances."""
    from oracle import torch_port as tp
    g = torch.Generator().manual_seed(6)
    S, Dz = 3 * N_PHONES, LATENT
    if cov == 'full':
        post = (torch.randn(S, Dz, generator=g), torch.ones(S, 1),
                torch.eye(Dz).repeat(S, 1, 1) / Dz, torch.full((S, 1), float(Dz)))
    else:
        post = (torch.randn(S, Dz, generator=g), torch.ones(S, 1), torch.ones(S, 1), torch.ones(S, Dz))
    trans = torch.full((S, S), -float('inf'))
    for st in range(S):
        trans[st, st] = np.log(.75)
        if st % 3 < 2:
            trans[st, st + 1] = np.log(.25)
        else:
            trans[st, ::3] = np.log(.25 / N_PHONES)
    init = torch.where(torch.arange(S) % 3 == 0, torch.tensor(np.log(1. / N_PHONES)),
                       torch.tensor(-float('inf'))).float()
    fin = torch.where(torch.arange(S) % 3 == 2, torch.tensor(np.log(.25)),
                      torch.tensor(-float('inf'))).float()
    rng = np.random.RandomState(4)
    nt = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    tp.vae_hmm_prior_path(torch.randn(50, Dz, generator=g), cov, post, init, fin, trans)
    t0 = time.perf_counter()
    frames = utts = 0
    while time.perf_counter() - t0 < budget_s:
        T = int(rng.randint(200, 401))
        tp.vae_hmm_prior_path(torch.randn(T, Dz, generator=g), cov, post, init, fin, trans)
        frames += T
        utts += 1
    dt = time.perf_counter() - t0
    cores = int(torch.get_num_threads())
    torch.set_num_threads(nt)
    return {'value': frames / dt, 'unit': 'frames/s', 'cores': cores, **host_cores(), 'kind': 'port',
            'sample': f'{utts} utterances ({frames} latent samples) through the prior hot path of '
                      f'config 4 ({cov} Gaussians, 64-d latent, 120 states): torch-CPU replay of the '
                      f'reference op sequence incl. autograd, {dt:.1f} s'}


def cpu_baseline_config5(signals, seqs, units, ploop, mean, var, epochs, n_comp, total, m):
    """The same stages on the host for a sample of `m` utterances, with the CPU restatements of
    the reference (oracle/features_oracle.py, graph_oracle.py, torch_port.hmm_elbo with the
    alignment graph's pdf ids, beer_oracle.best_path), projected to the corpus: stage time x
    (utterances / m) (x epochs for training)."""
    from oracle import beer_oracle as orc, features_oracle as fo, graph_oracle as go, torch_port as tp
    nutt = len(signals)
    pick = list(range(0, nutt, max(1, nutt // m)))[:m]
    t = {}
    t0 = time.perf_counter()
    sig_h = [signals[u].cpu().numpy() for u in pick]
    feats = [fo.extract(sg) for sg in sig_h]
    t['features'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    cgs = [go.compile_graph(go.alignment_graph(seqs[u], units, beer.graph.Graph)) for u in pick]
    t['alignment_graphs'] = time.perf_counter() - t0
    D = feats[0].shape[1]
    gen = torch.Generator().manual_seed(5)
    # the model's own groups (the recipe's JointModelSet: one MixtureSet per entry of hmm.yml)
    groups = []
    for ms in ploop.modelset.original_modelset.modelsets:
        KK, S_g, G_g = len(ms) * ms.n_comp_per_mixture, len(ms), ms.n_comp_per_mixture
        prior = (mean.cpu().float().repeat(KK, 1), torch.ones(KK, 1), torch.ones(KK, 1),
                 var.cpu().float().repeat(KK, 1))
        post = (prior[0] + .1 * torch.randn(KK, D, generator=gen) * var.cpu().float().sqrt(),) + prior[1:]
        w = torch.ones(S_g, G_g)
        groups.append((post, prior, w, w))
    nt = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    t0 = time.perf_counter()
    for f, cg in zip(feats, cgs):
        Xh = torch.from_numpy(f).float()
        with np.errstate(divide='ignore'):
            init, fin, trans = [torch.from_numpy(np.log(np.asarray(a, dtype=np.float32))) for a in cg[:3]]
        tp.hmm_elbo_groups(Xh, groups, init, fin, trans, total, trans_posteriors=False,
                           order=list(cg[3]))
    t['training_one_epoch'] = time.perf_counter() - t0
    t0 = time.perf_counter()
    for f, cg in zip(feats, cgs):
        with np.errstate(divide='ignore'):
            init, fin, trans = [np.log(np.asarray(a, dtype=np.float64)) for a in cg[:3]]
        pc = np.random.RandomState(0).randn(len(f), len(init))
        orc.best_path(pc, init, fin, trans)
    t['viterbi_align'] = time.perf_counter() - t0
    torch.set_num_threads(nt)
    scale = nutt / float(len(pick))
    proj = {'features': t['features'] * scale, 'alignment_graphs': t['alignment_graphs'] * scale,
            'training': t['training_one_epoch'] * scale * epochs, 'viterbi_align': t['viterbi_align'] * scale}
    wall = sum(proj.values())
    return {'value': total / wall, 'unit': 'frames/s', 'cores': int(min(16, os.cpu_count() or 1)),
            **host_cores(), 'kind': 'port', 'projected_wall_s': wall, 'projected_stages_s': proj,
            'sample_stages_s': t,
            'sample': f'{len(pick)} of the {nutt} utterances through the CPU restatements of the '
                      'reference stage by stage (features: numpy, one core; alignment graphs: the '
                      'reference\'s pure-Python builder + compile; training: torch replay of '
                      'evidence_lower_bound with the alignment graph, one epoch; Viterbi: numpy '
                      'best_path on the graph\'s states), projected to the corpus and the epochs'}


def cpu_baseline_config1(X, make):
    """Config 1 on the host: the port of the reference's op sequence for this model
    (oracle/torch_port.py: gmm_diag_iteration, pinned on the reference's golden G1), fp64, one
    thread, ~2 s of iterations.  `make()` builds the bench's model (its parameters are copied)."""
    from oracle import torch_port as tp
    as64 = lambda d: tuple(getattr(d.params, n).detach().cpu().double().clone()       # noqa: E731
                           for n in d._std_params_def)
    m = make()
    p0, p1 = list(m.bayesian_parameters())
    post, prior = as64(p0.posterior), as64(p0.prior)
    w_post, w_prior = as64(p1.posterior)[0], as64(p1.prior)[0]
    nt = torch.get_num_threads()
    torch.set_num_threads(1)
    for _ in range(3):
        tp.gmm_diag_iteration(X, post, prior, w_post, w_prior)
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < 2.:
        _, post, w_post = tp.gmm_diag_iteration(X, post, prior, w_post, w_prior)
        n += 1
    dt = (time.perf_counter() - t0) / n
    torch.set_num_threads(nt)
    return {'value': 1000 / dt, 'unit': 'frames/s', 'us_per_iteration': 1e6 * dt,
            'cores': 1, **host_cores(), 'kind': 'port',
            'sample': f'{n} iterations of the config-1 workload, torch-CPU replay of the reference '
                      'op sequence (oracle/torch_port.py: gmm_diag_iteration), one thread',
            'reference_calibration': 'the imported reference itself needs 0.160 s per iteration on '
                                     'the survey container (BASELINE.md section 2: overhead-bound '
                                     '-- Python objects, not arithmetic)'}
