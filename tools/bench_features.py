#!/usr/bin/env python
"""Feature front-end throughput: `beer features extract` default pipeline
(13 MFCC + energy + deltas + double deltas, 25 ms / 10 ms frames) over a batch
of synthetic 16 kHz utterances of U[2, 8] s, signals resident in HBM.
Also times the numpy oracle on a bounded sample on the host cores.

    python tools/bench_features.py --hours 2
"""

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import beer_amd as beer                                    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--hours', type=float, default=1., help='hours of audio in the batch')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--cpu-utts', type=int, default=40)
    args = ap.parse_args()
    rng = np.random.RandomState(6)
    dev = torch.device('cuda', 0)
    target = int(args.hours * 3600 * 16000)
    lengths = []
    while sum(lengths) < target:
        lengths.append(int(rng.randint(2 * 16000, 8 * 16000)))
    gen = torch.Generator(device=dev).manual_seed(6)
    sigs = [(torch.randn(n, generator=gen, device=dev) * 3000).to(torch.int16) for n in lengths]
    out = {'workload': f'MFCC+E+d+dd (39 dims), {len(sigs)} utterances, '
                       f'{sum(lengths) / 16000 / 3600:.2f} h of 16 kHz audio'}
    feats = beer.features.extract(sigs, as_numpy=False)
    nframes = sum(len(f) for f in feats)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        feats = beer.features.extract(sigs, as_numpy=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    out['gpu'] = {'ms': 1e3 * dt, 'frames_per_s': nframes / dt,
                  'x_real_time': sum(lengths) / 16000 / dt}
    import bench                                           # its cpu_baseline leg runs the oracle
    sample = [s.cpu().numpy() for s in sigs[:args.cpu_utts]]
    rate, ref = bench.cpu_baseline_features(sample)
    out['cpu_baseline'] = {'value': rate, 'unit': 'frames/s', 'cores': 1, 'kind': 'port',
                           'sample': f'{len(sample)} utterances, numpy oracle'}
    err = max(float(np.abs(r - f.cpu().numpy()).max()) for r, f in zip(ref, feats))
    out['max_abs_err_vs_oracle'] = err
    print(json.dumps(out))


if __name__ == '__main__':
    main()
