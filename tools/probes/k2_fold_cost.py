"""What a per-tile pass over the responsibilities in LDS costs K2: accx_kernel<true> (the state
posteriors of S = 2 states x 128 Gaussians folded into the packed tile between its arrival and its
first use) against accx_kernel<false> on the same 256 components, 1 M frames."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, beer_amd as beer
from beer_amd import _hip, kernels
dev = torch.device('cuda:0')
T, D, K = 1_000_000, 40, 256
torch.manual_seed(0)
X = torch.randn(T, D, device=dev)
ns = beer.NormalSet.create(torch.zeros(D), torch.eye(D), size=K, prior_strength=1., noise_std=1., cov_type='full')
ms = beer.MixtureSet.create(2, ns).float().to(dev)
E, lw = ns.means_precisions.natural_form(), ms._log_weights()
st = beer.FrameStats(X, 'full')
ln, packed = kernels.mixtureset_estep_packed(st, E, lw, 2, 128, 'full')
sr = torch.ones(T, 2, device=dev)
def t(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for rep in range(3):
    a = t(lambda: kernels.normal_accumulate(st, packed, sr, 2, 128, 'full'))
    mix = beer.Mixture.create(ns).float().to(dev)
    ln1, p1 = kernels.mixture_estep_packed(st, E, mix._log_weights().view(1, K), K, 'full')
    b = t(lambda: kernels.normal_accumulate(st, p1, None, 1, K, 'full'))
    k1s = t(lambda: kernels.mixtureset_estep_packed(st, E, lw, 2, 128, 'full'))
    k1 = t(lambda: kernels.mixture_estep_packed(st, E, mix._log_weights().view(1, K), K, 'full'))
    print(f'K2 with fold (SR) {a:.3f} ms   K2 plain {b:.3f} ms   K1 sets {k1s:.3f}  K1 mixture {k1:.3f}', flush=True)
