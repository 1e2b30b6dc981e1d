"""cProfile of the host side of one config-3 iteration (device idle in between: after a sync)."""
import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench, beer_amd as beer
from beer_amd.distributed import all_reduce_elbo
dev = torch.device('cuda:0')
lengths = bench.hmm_corpus(10_000_000)
X = torch.randn(sum(lengths), bench.D, device=dev)
ploop = bench.make_phone_loop('diagonal', dev)
optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.)
def step():
    optim.init_step()
    elbo = beer.accumulate_elbo(ploop, (X, lengths), datasize=sum(lengths))
    elbo, _ = all_reduce_elbo(elbo, ploop, len(lengths))
    elbo.backward()
    optim.step()
for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
