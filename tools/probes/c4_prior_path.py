"""The prior hot path of config 4 alone (bench.py: run_vae's `prior_path`), n times, for a
kernel trace:  python tools/probes/c4_prior_path.py [full|diagonal] [n] [dense]"""
import sys
import time
import torch
sys.path.insert(0, '.')
import bench
import beer_amd as beer

cov = sys.argv[1] if len(sys.argv) > 1 else 'full'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 6
dense = len(sys.argv) > 3
dev = torch.device('cuda')
lengths = bench.hmm_corpus(1_000_000)
total = sum(lengths)
torch.manual_seed(4)
prior = bench.make_phone_loop(cov, dev, dim=bench.LATENT, n_comp=1)
Z = torch.randn(total, bench.LATENT, device=dev)
for i in range(n + 2):
    if i == 2:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
    z = Z.clone().requires_grad_(True)
    stats = beer.kernels.differentiable_stats(z, cov, 1) if dense else beer.kernels.sample_stats(z, cov)
    exp_llh = prior.expected_log_likelihood(stats, utt_lengths=lengths)
    exp_llh.sum().backward()
    acc = prior.accumulate(stats.detach())
    prior.clear_cache()
torch.cuda.synchronize()
print(f'{cov} {"dense" if dense else "one-sample"} route: {1e3 * (time.perf_counter() - t0) / n:.2f} ms per pass of {total} frames')
