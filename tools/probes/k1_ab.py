"""K1 (packed full-covariance E-step) with the parameters staged through LDS per workgroup
(BEER_OPT_K1_LDS = 1) against per-wave streaming (0): bit-identical outputs, time per launch.
    python tools/probes/k1_ab.py [K] [frames]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, beer_amd as beer
from beer_amd import _hip, kernels
K = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
D = 40
dev = torch.device('cuda:0')
torch.manual_seed(0)
X = torch.randn(T, D, device=dev)
ns = beer.NormalSet.create(torch.zeros(D), torch.eye(D), size=K, prior_strength=1., noise_std=1., cov_type='full')
mix = beer.Mixture.create(ns).to(dev)
E, lw = ns.means_precisions.natural_form(), mix._log_weights().view(1, K)
st = beer.FrameStats(X, 'full')
res = {}
for mode in (0, 1, 0, 1):
    _hip.set_option('k1_lds', mode)
    for _ in range(2):
        ln, packed = kernels.mixture_estep_packed(st, E, lw, K, 'full')
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ln, packed = kernels.mixture_estep_packed(st, E, lw, K, 'full')
    b.record()
    torch.cuda.synchronize()
    r = packed.unpack()
    print(f'k1_lds={mode}: {a.elapsed_time(b) / 10:.3f} ms per call; ln sum {float(ln.double().sum()):.9e}')
    res.setdefault(mode, (ln.clone(), r.clone()))
print('bit-identical log-normalisers:', torch.equal(res[0][0], res[1][0]), ' responsibilities:', torch.equal(res[0][1], res[1][1]))
