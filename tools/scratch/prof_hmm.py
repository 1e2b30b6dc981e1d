import sys, cProfile, pstats, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import beer_amd as beer
from bench_hmm import build
dev = torch.device('cuda', 0)
rng = np.random.RandomState(2)
lengths = []
while sum(lengths) < 1000000: lengths.append(int(rng.randint(200, 401)))
X = torch.randn(sum(lengths), 40, device=dev)
ploop, units = build(40, 16, 40, 'diagonal', dev, torch.float32)
optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.)
def run():
    optim.init_step()
    elbo = beer.accumulate_elbo(ploop, (X, lengths), datasize=sum(lengths))
    elbo.backward(); optim.step()
    torch.cuda.synchronize()
run(); run()
pr = cProfile.Profile(); pr.enable(); run(); run(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
