"""accfp_kernel (option accfi_pipe = 1) against accfi_kernel: same statistics up to the order of the fp64 sums."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import beer_amd as beer
from beer_amd import _hip, kernels
DEV = 'cuda'
for (T, D, S, G, cov) in ((100003, 40, 120, 16, 'diagonal'), (33001, 20, 9, 16, 'diagonal'), (20000, 40, 37, 8, 'diagonal'), (17000, 12, 6, 32, 'isotropic')):
    torch.manual_seed(1)
    X = torch.randn(T, D, device=DEV) * 2.
    K = S * G
    ns = beer.NormalSet.create(torch.zeros(D), torch.ones(D) if cov == 'diagonal' else torch.ones(1), size=K, prior_strength=1., noise_std=1.5, cov_type=cov)
    E = ns.means_precisions.natural_form().float().to(DEV)
    lw = torch.log_softmax(torch.randn(S, G, device=DEV), dim=1)
    st = beer.FrameStats(X, cov)
    ln, _ = kernels.mixtureset_estep(st, E, lw, S, G, cov, want_resps=False)
    sr = torch.softmax(torch.randn(T, S, device=DEV), dim=1)
    out = {}
    for opt in (0, 1):
        _hip.set_option('accfi_pipe', opt)
        out[opt] = kernels.mixtureset_accumulate_fused(st, E, lw, ln, sr, S, G, cov).clone()
        out[(opt, 'nosr')] = kernels.mixtureset_accumulate_fused(st, E, lw, ln, None, S, G, cov).clone()
    _hip.set_option('accfi_pipe', 0)
    sc = float(out[0].abs().max())
    print((T, D, S, G, cov), 'max rel diff', float((out[0] - out[1]).abs().max()) / sc, 'no sr', float((out[(0,'nosr')] - out[(1,'nosr')]).abs().max()) / float(out[(0,'nosr')].abs().max()),
          'counts', float((out[0][:, -1] - out[1][:, -1]).abs().max() / out[0][:, -1].abs().max()), 'nan', bool(torch.isnan(out[1]).any()))
