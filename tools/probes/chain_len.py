"""Statistics error of the bf16x3 accumulation alone (fp64 responsibilities given) against the
fp64 kernels, as a function of the MFMA chain length (BEER_AX_MAXFRAMES, read once per process:
run this script once per value)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import beer_amd as beer
from beer_amd import kernels
DEV = 'cuda'
K, D, T = 256, 40, 1 << 20
rng = np.random.RandomState(3)
means = rng.randn(K, D) * 2
Xn = (means[rng.randint(0, K, T)] + rng.randn(T, D)).astype(np.float32)
X = torch.from_numpy(Xn).to(DEV)
torch.manual_seed(7)
ns = beer.NormalSet.create(X.mean(0).cpu(), torch.diag(X.var(0).cpu()), size=K, prior_strength=1.,
                           noise_std=1., cov_type='full')
model = beer.Mixture.create(ns, prior_strength=1.).to(DEV)
E, lw = ns.means_precisions.natural_form(), model._log_weights().view(1, K)
st64, st32 = beer.FrameStats(X.double(), 'full'), beer.FrameStats(X, 'full')
_, r64 = kernels.mixtureset_estep(st64, E.double(), lw.double(), 1, K, 'full')
acc64 = kernels.normal_accumulate(st64, r64, None, 1, K, 'full')
acc = kernels.normal_accumulate(st32, kernels.pack_resps(st32, r64.float(), None, 1, K), None, 1, K, 'full')
e = (acc - acc64).abs()
cnt = -2 * acc[:, -2]; cnt64 = -2 * acc64[:, -2]
diag = torch.arange(D, device=DEV) * (D + 1) + D
print(os.environ.get('BEER_AX_MAXFRAMES'), 'all', float(e.max() / acc64.abs().max()),
      'counts max rel', float(((cnt - cnt64).abs() / cnt64).max()), 'counts mean rel (bias)', float(((cnt - cnt64) / cnt64).mean()),
      'squares mean rel (bias)', float(((acc[:, diag] - acc64[:, diag]) / acc64[:, diag]).mean()))
