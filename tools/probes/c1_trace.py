"""Config 1 (K = 8, D = 2, 1000 fp64 frames) as a captured iteration: which kernels a replay runs."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, beer_amd as beer
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
X = torch.randn(1000, 2, dtype=torch.float64, generator=g)
torch.manual_seed(0)
ns = beer.NormalSet.create(X.mean(0), X.var(0), size=8, prior_strength=1., noise_std=1., cov_type='diagonal')
m = beer.Mixture.create(ns, prior_strength=1.).double().to(dev)
it = beer.CapturedIteration(m, beer.VBConjugateOptimizer(m.mean_field_factorization(), 1.), X.to(dev))
for _ in range(60):
    it()
torch.cuda.synchronize()
print(it.mode)
