import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import beer_amd as beer
from beer_amd import kernels, _hip
DEV='cuda'
for cov, K, D in (('full', 64, 64), ('full', 64, 68), ('full', 64, 80), ('diagonal', 64, 80)):
    T = 17000
    torch.manual_seed(1)
    X = torch.randn(T, D, dtype=torch.float64, device=DEV) * 1.5
    var = torch.ones(D) if cov != 'full' else torch.eye(D)
    ns = beer.NormalSet.create(torch.zeros(D), var, size=K, prior_strength=1., noise_std=1., cov_type=cov)
    model = beer.Mixture.create(ns).double().to(DEV)
    E64, lw64 = ns.means_precisions.natural_form(), model._log_weights().view(1, K)
    st64, st32 = beer.FrameStats(X, cov), beer.FrameStats(X.float(), cov)
    ln64, r64 = kernels.mixtureset_estep(st64, E64, lw64, 1, K, cov)
    acc64 = kernels.normal_accumulate(st64, r64, None, 1, K, cov)
    ln, packed = kernels.mixture_estep_packed(st32, E64.float(), lw64.float(), K, cov)
    r = packed.unpack()
    acc = kernels.normal_accumulate(st32, packed, None, 1, K, cov)
    accr = kernels.normal_accumulate(st32, kernels.pack_resps(st32, r64.float(), None, 1, K), None, 1, K, cov)
    e = (acc - acc64).abs(); er = (accr - acc64).abs()
    print(cov, K, D, 'ln err', float((ln.double()-ln64).abs().max()), 'r err', float((r.double()-r64).abs().max()),
          'acc err', float(e.max()/acc64.abs().max()), 'acc(given r) err', float(er.max()/acc64.abs().max()),
          'worst col', int(er.max(0)[1].argmax()) if False else int(er.amax(0).argmax()), 'of', acc.shape[1])
