"""The two groups of the recipes' JointModelSet (5 states x 10 Gaussians, 120 x 4) at D = 42 over
1.07 M frames: log-normalisers + fused accumulation per group, over BEER_OPT_ACCF_ROUNDS."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, beer_amd as beer
from beer_amd import _hip, kernels
dev = torch.device('cuda:0')
T, D = 1_072_915, int(sys.argv[1]) if len(sys.argv) > 1 else 42
torch.manual_seed(0)
X = torch.randn(T, D, device=dev)
images = beer.FrameImages(X)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for S, G in ((5, 10), (120, 4), (125, 4), (120, 16)):
    ns = beer.NormalSet.create(torch.zeros(D), torch.ones(D), size=S * G, prior_strength=1., noise_std=1., cov_type='diagonal')
    ms = beer.MixtureSet.create(S, ns).float().to(dev)
    E, lw = ns.means_precisions.natural_form(), ms._log_weights()
    st = beer.FrameStats(X, 'diagonal', images=images)
    ln, _ = kernels.mixtureset_estep(st, E, lw, S, G, 'diagonal', want_resps=False)
    sr = torch.softmax(torch.randn(T, S, device=dev), dim=1)
    row = [f'S={S} G={G}: estep {t(lambda: kernels.mixtureset_estep(st, E, lw, S, G, "diagonal", want_resps=False)):.3f} ms; accumulate']
    for rounds in (1, 2, 3, 6, 12):
        _hip.set_option('accf_rounds', rounds)
        row.append(f'rounds {rounds}: {t(lambda: kernels.mixtureset_accumulate_fused(st, E, lw, ln, sr, S, G, "diagonal")):.3f}')
    _hip.set_option('accf_rounds', 6)
    print('  '.join(row), flush=True)
