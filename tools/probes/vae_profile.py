"""Where a config-4 VAE minibatch step spends its time outside the prior's kernels (torch profiler)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench, beer_amd as beer
dev = torch.device('cuda:0')
cov = sys.argv[1] if len(sys.argv) > 1 else 'diagonal'
lens = bench.hmm_corpus(1_000_000)
T = sum(lens)
X = torch.randn(T, bench.D, device=dev)
torch.manual_seed(4)
prior = bench.make_phone_loop(cov, dev, dim=bench.LATENT, n_comp=1)
vae = beer.VAE(prior, beer.nnet.ResidualFeedForwardNet(bench.D, 2, 128),
               beer.nnet.ResidualFeedForwardNet(bench.LATENT, 2, 128)).to(dev)
cjg = beer.VBConjugateOptimizer(vae.mean_field_factorization(), lrate=.1)
optim = beer.VBOptimizer(cjg, torch.optim.Adam(vae.parameters(), lr=1e-3))
def step():
    optim.init_step()
    elbo = beer.accumulate_elbo(vae, (X, lens), datasize=5 * T)
    elbo.backward()
    optim.step()
for _ in range(2): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): step()
torch.cuda.synchronize()
print('ms per step', (time.perf_counter() - t0) / 3 * 1e3)
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by='cuda_time_total', row_limit=30, max_name_column_width=60))
