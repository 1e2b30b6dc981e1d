#!/usr/bin/env python
"""Config-4 shaped workload: HMM-VAE.  D = 40 frames, residual feed-forward
encoder / decoder, 64-dimensional Normal latent, phone-loop HMM prior
(40 phones x 3 states, one Gaussian per state) over ragged utterances of
U[200, 400] frames.  One step = ELBO (encoder, sampling, statistics-in prior
E-step, decoder) + backward to the networks + sufficient-statistic
accumulation + natural-gradient / Adam update on one minibatch.

Prints one JSON line: frames/s of the whole step and of the prior's hot path
alone (statistics -> llh -> forward-backward -> gradient -> accumulation).

    python tools/bench_vae.py --frames 200000 --cov diagonal --nsamples 1
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
from tools.bench_hmm import unit                            # noqa: E402


def build_prior(n_phones, Dz, cov, device):
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
    torch.manual_seed(4)
    ns = beer.NormalSet.create(torch.zeros(Dz), torch.ones(Dz), size=3 * n_phones,
                               prior_strength=1., noise_std=1., cov_type=cov)
    start_pdf = {p: 3 * p for p in units}
    end_pdf = {p: 3 * p + 2 for p in units}
    return beer.PhoneLoop.create(graph.compile(), start_pdf, end_pdf, ns).to(device)


def timed(fn, steps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--frames', type=int, default=200_000, help='frames per minibatch')
    ap.add_argument('--phones', type=int, default=40)
    ap.add_argument('--dim', type=int, default=40)
    ap.add_argument('--latent', type=int, default=64)
    ap.add_argument('--cov', default='diagonal')
    ap.add_argument('--nsamples', type=int, default=1)
    ap.add_argument('--width', type=int, default=128)
    ap.add_argument('--steps', type=int, default=5)
    args = ap.parse_args()
    dev = torch.device('cuda', 0)
    rng = np.random.RandomState(4)
    lengths = []
    while sum(lengths) < args.frames:
        lengths.append(int(rng.randint(200, 401)))
    total = sum(lengths)
    gen = torch.Generator(device=dev).manual_seed(4)
    X = torch.randn(total, args.dim, generator=gen, device=dev)
    prior = build_prior(args.phones, args.latent, args.cov, dev)
    vae = beer.VAE(prior, beer.nnet.ResidualFeedForwardNet(args.dim, 2, args.width),
                   beer.nnet.ResidualFeedForwardNet(args.latent, 2, args.width)).to(dev)
    cjg = beer.VBConjugateOptimizer(vae.mean_field_factorization(), lrate=.1)
    optim = beer.VBOptimizer(cjg, torch.optim.Adam(vae.parameters(), lr=1e-3))
    out = {'workload': f'HMM-VAE: D={args.dim}, latent {args.latent} ({args.cov} prior '
                       f'Gaussians), phone loop {args.phones}x3 states, nsamples='
                       f'{args.nsamples}, {len(lengths)} utts, {total} frames / minibatch'}

    def step():
        optim.init_step()
        elbo = beer.accumulate_elbo(vae, (X, lengths), datasize=25 * total,
                                    nsamples=args.nsamples)
        elbo.backward()
        optim.step()
        return elbo

    dt, elbo = timed(step, args.steps)
    out['vae_step'] = {'ms_per_step': 1e3 * dt, 'frames_per_s': total / dt,
                       'elbo_per_frame': float(elbo) / (25 * total)}

    # the prior's hot path alone, on fixed samples of the latent variable
    Z = torch.randn(total * args.nsamples, args.latent, generator=gen, device=dev)

    def prior_path():
        z = Z.clone().requires_grad_(True)
        stats = beer.kernels.differentiable_stats(z, args.cov, args.nsamples)
        exp_llh = prior.expected_log_likelihood(stats, utt_lengths=lengths)
        exp_llh.sum().backward()
        acc = prior.accumulate(stats.detach())
        prior.clear_cache()
        return acc

    dt, _ = timed(prior_path, args.steps)
    out['prior_hot_path'] = {'ms': 1e3 * dt, 'frames_per_s': total / dt}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
