#!/usr/bin/env python
"""Build-container check of the CPU baselines `bench.py` reports.

`bench.py`'s `cpu_baseline` legs time oracle/torch_port.py (kind "port": the
reference's op sequence replayed with torch CPU ops) because the reference
itself cannot travel to the GPU box.  This script runs in the build container,
where /root/reference is importable, and times BOTH on the same inputs:

  config 2 (GMM, K = 256 full covariance, D = 40, float32, 8192-frame
  utterances): `beer.evidence_lower_bound` + `backward` + optimizer step of the
  imported reference against `torch_port.gmm_iteration`, at 1 and 8 threads.

  config 3 shape (HMM, 120 states in a phone-loop-like graph, 16 diagonal Gaussians per
  state, D = 40, float32, 300-frame utterances): `beer.evidence_lower_bound(hmm, x)` per
  utterance against `torch_port.hmm_elbo`, at 1 and 8 threads.

It also compares the two ELBO values (they are the same arithmetic).  Output:
one JSON object, committed as profiles/r02_cpu_baseline_crosscheck.json.

    python tools/ref_timing_check.py > profiles/r02_cpu_baseline_crosscheck.json
"""

import json
import os
import sys
import time

sys.dont_write_bytecode = True     # (importing the reference must not drop __pycache__ into it)

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get('BEER_REFERENCE', '/root/reference')


def _best_of(fn, reps):
    times = []
    for _ in range(reps):
        t = time.perf_counter()
        out = fn()
        times.append(time.perf_counter() - t)
    return min(times), out


def gmm_case(beer, tp, threads):
    K, D, chunk, nutt = 256, 40, 8192, 2
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(3)
    n = nutt * chunk
    means = torch.randn(K, D, generator=g) * 2
    X = means[torch.randint(0, K, (n,), generator=g)] + torch.randn(n, D, generator=g)
    mean, cov = X.mean(0), torch.cov(X.t())
    # the reference model
    ns = beer.NormalSet.create(mean, cov, size=K, prior_strength=1., noise_std=1.,
                               cov_type='full')
    model = beer.Mixture.create(ns, prior_strength=1.).float()
    mp = ns.means_precisions
    post = tuple(getattr(mp.posterior.params, a).clone() for a in
                 ('mean', 'scale', 'scale_matrix', 'dof'))
    prior = tuple(getattr(mp.prior.params, a).clone() for a in
                  ('mean', 'scale', 'scale_matrix', 'dof'))
    wp = model.categorical.weights
    w_post = wp.posterior.params.concentrations.clone()
    w_prior = wp.prior.params.concentrations.clone()

    def reference():
        optim = beer.VBConjugateOptimizer(model.mean_field_factorization(), lrate=1.)
        optim.init_step()
        elbo = beer.evidence_lower_bound(datasize=n)
        for u in range(nutt):
            elbo += beer.evidence_lower_bound(model, X[u * chunk:(u + 1) * chunk], datasize=n)
        elbo.backward()
        optim.step()
        return float(elbo)

    def port():
        return tp.gmm_iteration(X, post, prior, w_post, w_prior, chunk)

    # one iteration each from the same starting point: the reference updates its
    # model in place, so it goes second on the values and is re-created for timing
    v_port = port()
    v_port = float(v_port[0] if isinstance(v_port, (tuple, list)) else v_port)
    t_ref, v_ref = _best_of(reference, 1)
    t_port, _ = _best_of(port, 2)
    return {'workload': f'GMM K={K} full, D={D}, float32, {nutt} x {chunk} frames, 1 VB iteration',
            'threads': threads, 'reference_s': t_ref, 'port_s': t_port,
            'reference_frames_per_s': n / t_ref, 'port_frames_per_s': n / t_port,
            'port_over_reference_speed': t_ref / t_port,
            'elbo_reference': v_ref, 'elbo_port': v_port,
            'elbo_rel_diff': abs(v_ref - v_port) / abs(v_ref)}


def hmm_case(beer, tp, threads):
    'Config-3 shaped HMM: 120 states in a loop, 16 diagonal Gaussians per state, D = 40.'
    S, G, D, nutt = 120, 16, 40, 3
    torch.set_num_threads(threads)
    g = torch.Generator().manual_seed(5)
    graph = beer.graph.Graph()
    start, end = graph.add_state(), graph.add_state()
    graph.start_state, graph.end_state = start, end
    states = [graph.add_state(pdf_id=s) for s in range(S)]
    for s in range(S):
        graph.add_arc(states[s], states[s])
        if s % 3 < 2:
            graph.add_arc(states[s], states[s + 1])
        else:
            graph.add_arc(states[s], end)
            for t in range(0, S, 3):
                graph.add_arc(states[s], states[t])
        if s % 3 == 0:
            graph.add_arc(start, states[s])
    graph.normalize()
    cgraph = graph.compile()
    ns = beer.NormalSet.create(torch.zeros(D), torch.ones(D), size=S * G, prior_strength=1.,
                               noise_std=1., cov_type='diagonal')
    ms = beer.MixtureSet.create(S, ns, prior_strength=1.)
    hmm = beer.HMM.create(cgraph, ms).float()
    mp = ns.means_precisions
    names = ('mean', 'scale', 'shape', 'rates')
    post = tuple(getattr(mp.posterior.params, a).clone() for a in names)
    prior = tuple(getattr(mp.prior.params, a).clone() for a in names)
    wp = ms.categoricalset.weights
    w_post = wp.posterior.params.concentrations.clone()
    w_prior = wp.prior.params.concentrations.clone()
    init, fin, trans = (cgraph.init_log_probs.float(), cgraph.final_log_probs.float(),
                        cgraph.trans_log_probs.float())
    utts = [torch.randn(300, D, generator=g) for _ in range(nutt)]
    N = 10_000_000

    def reference():
        return sum(float(beer.evidence_lower_bound(hmm, x, datasize=N)) for x in utts)

    def port():
        return sum(float(tp.hmm_elbo(x, post, prior, w_post, w_prior, init, fin, trans, N)[0])
                   for x in utts)

    t_port, v_port = _best_of(port, 2)
    t_ref, v_ref = _best_of(reference, 2)
    n = 300 * nutt
    return {'workload': f'HMM {S} states x {G} diagonal Gaussians, D={D}, float32, {nutt} x 300 '
                        'frames, E-step + statistics (no update)',
            'threads': threads, 'reference_s': t_ref, 'port_s': t_port,
            'reference_frames_per_s': n / t_ref, 'port_frames_per_s': n / t_port,
            'port_over_reference_speed': t_ref / t_port,
            'elbo_reference': v_ref, 'elbo_port': v_port,
            'elbo_rel_diff': abs(v_ref - v_port) / abs(v_ref)}


def main():
    sys.path.insert(0, REF)
    import beer                                             # the reference itself
    sys.path.pop(0)
    from oracle import torch_port as tp
    out = {'host': {'cpus': os.cpu_count(), 'torch': torch.__version__},
           'what': 'imported reference (/root/reference) vs oracle/torch_port.py on the same '
                   'inputs, build container; bench.py reports the port on the GPU box',
           'cases': []}
    for threads in (1, 8):
        out['cases'].append(gmm_case(beer, tp, threads))
    for threads in (1, 8):
        out['cases'].append(hmm_case(beer, tp, threads))
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
