#!/usr/bin/env python
"""How far the fp16-split E-step + accumulation are from the fp64 kernels on mixtures
with few frames per component (K = 512 .. 768, D = 16 / 40, 17 000 frames), next to the
generic float32 kernels: the numbers quoted in DESIGN.md section 8.  Needs an MI355X.

    python tools/split_precision_probe.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import beer_amd as beer
from beer_amd import kernels
DEV = torch.device('cuda')
def run(cov, K, D, splits):
    T = 17000
    rng = np.random.RandomState(K + D)
    means = rng.randn(K, D) * 2
    Xn = (means[rng.randint(0, K, T)] + rng.randn(T, D) * (1 + .3 * rng.rand(D))).astype(np.float32)
    X = torch.from_numpy(Xn).to(DEV)
    torch.manual_seed(5)
    var = torch.from_numpy(Xn).var(0) if cov != 'full' else torch.diag(torch.from_numpy(Xn).var(0))
    ns = beer.NormalSet.create(torch.from_numpy(Xn).mean(0), var, size=K, prior_strength=1., noise_std=1., cov_type=cov)
    model = beer.Mixture.create(ns).to(DEV)
    E = ns.means_precisions.natural_form(); lw = model._log_weights().view(1, K)
    st32 = beer.FrameStats(X, cov); st64 = beer.FrameStats(X.double(), cov)
    ln64, r64 = kernels.mixtureset_estep(st64, E.double(), lw.double(), 1, K, cov)
    acc64 = kernels.normal_accumulate(st64, r64, None, K, 1, cov)
    ln32, r32 = kernels.mixtureset_estep(st32, E, lw, 1, K, cov)
    acc32 = kernels.normal_accumulate(st32, r32, None, K, 1, cov)
    sc = float(acc64.abs().max())
    print(cov, K, D, 'generic fp32: ln', float((ln32.double()-ln64).abs().max()), 'acc', float((acc32-acc64).abs().max())/sc)
    for sp in splits:
        ln, wr = kernels.wide_mixture_estep(st32, E, lw, K, cov, sp)
        acc = kernels.normal_accumulate(st32, wr, None, K, 1, cov)
        lse64, _ = kernels.mixtureset_estep(st64, E.double(), lw.double().reshape(sp), sp[0], sp[1], cov)
        print('   split', sp, 'ln', float((ln.double()-ln64).abs().max()), 'block lse', float((wr.block_lse.double()-lse64).abs().max()),
              'acc', float((acc-acc64).abs().max())/sc, 'N', float((acc[:, -1]-acc64[:, -1]).abs().max())/float(acc64[:, -1].abs().max()))
run('isotropic', 768, 16, [(3, 256), (6, 128), (12, 64)])
run('diagonal', 768, 16, [(3, 256), (6, 128)])
run('isotropic', 768, 40, [(3, 256), (6, 128)])
run('isotropic', 512, 16, [(2, 256), (8, 64)])
