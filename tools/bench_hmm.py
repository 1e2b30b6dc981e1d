#!/usr/bin/env python
"""Config-3 shaped workload: monophone phone-loop HMM, 40 phones x 3 states,
G Gaussians per state, D = 40, ragged utterances of U[200, 400] frames.
Prints one JSON line with frames/s per VB iteration for the free phone loop
and (optionally) per-utterance alignment graphs.

    python tools/bench_hmm.py --frames 1000000 --cov diagonal --ncomp 16
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import beer_amd as beer                                    # noqa: E402

TOPO = [(0, 1, 1.), (1, 1, .75), (1, 2, .25), (2, 2, .75), (2, 3, .25), (3, 3, .75),
        (3, 4, .25)]


def unit(start_pdf_id):
    g = beer.graph.Graph()
    for sid in range(5):
        g.add_state(pdf_id=None if sid in (0, 4) else start_pdf_id + sid - 1)
    g.start_state, g.end_state = 0, 4
    for arc in TOPO:
        g.add_arc(*arc)
    return g, start_pdf_id + 3


def build(n_phones, ncomp, D, cov, device, dtype):
    units, pdf = {}, 0
    for p in range(n_phones):
        units[p], pdf = unit(pdf)
    graph = beer.graph.Graph()
    graph.start_state, graph.end_state = graph.add_state(), graph.add_state()
    pivot = graph.add_state()
    u2s = {p: graph.add_state() for p in units}
    graph.add_arc(graph.start_state, pivot)
    graph.add_arc(pivot, graph.end_state)
    for p in units:
        graph.add_arc(pivot, u2s[p])
        graph.add_arc(u2s[p], pivot)
    graph.normalize()
    for p, hmm in units.items():
        graph.replace_state(u2s[p], hmm)
    graph.normalize()
    cgraph = graph.compile()
    start_pdf = {p: 3 * p for p in units}
    end_pdf = {p: 3 * p + 2 for p in units}
    torch.manual_seed(3)
    S = 3 * n_phones
    ns = beer.NormalSet.create(torch.zeros(D), torch.ones(D), size=S * ncomp, prior_strength=1.,
                               noise_std=1., cov_type=cov)
    emissions = beer.JointModelSet([beer.MixtureSet.create(S, ns, prior_strength=1.)])
    ploop = beer.PhoneLoop.create(cgraph, start_pdf, end_pdf, emissions)
    ploop = ploop.double() if dtype == torch.float64 else ploop.float()
    return ploop.to(device), units


def ali_graph(seq, units):
    graph = beer.graph.Graph()
    graph.start_state = graph.add_state()
    last, states = graph.start_state, []
    for p in seq:
        s = graph.add_state()
        states.append(s)
        graph.add_arc(last, s)
        last = s
    graph.end_state = graph.add_state()
    graph.add_arc(last, graph.end_state)
    for s, p in zip(states, seq):
        graph.replace_state(s, units[p])
    graph.normalize()
    return graph.compile()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=1_000_000)
    ap.add_argument('--phones', type=int, default=40)
    ap.add_argument('--ncomp', type=int, default=16)
    ap.add_argument('--dim', type=int, default=40)
    ap.add_argument('--cov', default='diagonal')
    ap.add_argument('--steps', type=int, default=3)
    ap.add_argument('--ali-utts', type=int, default=0, help='also time N utterances with '
                    'alignment graphs')
    ap.add_argument('--gc', default='freeze', choices=['freeze', 'default', 'off'],
                    help='Python garbage collector during the timed loops: freeze the set-up '
                         'objects (default; a generation-2 collection over them costs 25-40 ms '
                         'every few iterations), leave it alone, or switch it off')
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    rng = np.random.RandomState(2)
    lengths = []
    while sum(lengths) < args.frames:
        lengths.append(int(rng.randint(200, 401)))
    total = sum(lengths)
    g = torch.Generator(device=dev).manual_seed(2)
    X = torch.randn(total, args.dim, generator=g, device=dev)
    ploop, units = build(args.phones, args.ncomp, args.dim, args.cov, dev, torch.float32)
    optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.)
    out = {'workload': f'phone loop {args.phones}x3 states, G={args.ncomp} {args.cov}, '
                       f'D={args.dim}, {len(lengths)} utts, {total} frames'}

    def run(graphs, X_, lengths_):
        optim.init_step()
        elbo = beer.accumulate_elbo(ploop, (X_, lengths_), datasize=sum(lengths_),
                                    inference_graphs=graphs)
        elbo.backward()
        optim.step()
        return elbo

    for _ in range(2):                                  # allocator + workspaces settle
        run(None, X, lengths)
    import gc
    if args.gc == 'freeze':
        gc.collect()
        gc.freeze()
    elif args.gc == 'off':
        gc.disable()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        elbo = run(None, X, lengths)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    out['free_loop'] = {'ms_per_iter': 1e3 * dt, 'frames_per_s': total / dt,
                        'elbo_per_frame': float(elbo) / (len(lengths) * total)}
    # Viterbi decoding of the whole shard (HMM.decode, batched)
    beer.decode_batch(ploop, (X, lengths))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        paths = beer.decode_batch(ploop, (X, lengths))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    out['viterbi_decode'] = {'ms': 1e3 * dt, 'frames_per_s': total / dt}
    if args.ali_utts:
        n = min(args.ali_utts, len(lengths))
        sub = lengths[:n]
        seqs = [list(rng.randint(0, args.phones, max(2, T // 30))) for T in sub]
        t0 = time.perf_counter()
        gset = beer.graph.compile_alignments(seqs, units)
        graphs = list(gset)
        t1 = time.perf_counter()
        gset.device_image(torch.float32)
        torch.cuda.synchronize()
        out['ali_graph_build'] = {'utts': n, 'native_compile_s': t1 - t0,
                                  'device_image_s': time.perf_counter() - t1,
                                  'states': int(gset.state_off[-1]), 'arcs': int(gset.arc_off[-1])}
        # the reference's way: per-utterance pure-Python builder + compile, on a sample
        import bench                                       # its cpu_baseline leg runs the oracle
        m = min(n, 50)
        out['ali_graph_build']['cpu_baseline'] = {
            'value': bench.cpu_baseline_graph_compile(seqs[:m], units, beer.graph.Graph),
            'unit': 's/utterance', 'cores': 1, 'kind': 'port', 'sample': f'{m} utterances'}
        Xs = X[:sum(sub)]
        for _ in range(2):
            run(graphs, Xs, sub)
        if args.gc == 'freeze':
            gc.collect()
            gc.freeze()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            run(graphs, Xs, sub)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / args.steps
        out['alignment_graphs'] = {'utts': n, 'frames': sum(sub), 'ms_per_iter': 1e3 * dt,
                                   'frames_per_s': sum(sub) / dt}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
