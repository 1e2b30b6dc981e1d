"""Forward-backward accuracy / speed probe (GPU): the one-wave kernel on a phone loop of
config 3's shape, float32 against its own float64 run on the same (float32-valued)
per-state log-likelihoods.

    python tools/probes/fb_err.py [n_utts] [spread]
"""
import sys, os, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from beer_amd import hmm_kernels as hk

n_utts = int(sys.argv[1]) if len(sys.argv) > 1 else 400
spread = float(sys.argv[2]) if len(sys.argv) > 2 else 5.
dev = torch.device('cuda:0')
ploop = bench.make_phone_loop('diagonal', dev)
graph = ploop.graph
rng = np.random.RandomState(0)
lengths = [int(rng.randint(200, 401)) for _ in range(n_utts)]
T = sum(lengths)
S = graph.n_states
g = torch.Generator(device=dev).manual_seed(1)
pc32 = (torch.randn(T, S, generator=g, device=dev) * spread - 60.).float()
out = {}
for dt in (torch.float32, torch.float64):
    gr = graph if dt == torch.float32 else ploop.double().graph
    batch = hk.HmmBatch([gr], [0] * n_utts, lengths, dt)
    pc = pc32.to(dt)
    utt = torch.zeros(n_utts, dtype=torch.float64, device=dev)
    sr, g0, flow = hk.posteriors_fused(batch, pc, 1., want_counts=True, utt_llh=utt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        hk.posteriors_fused(batch, pc, 1., want_counts=True, utt_llh=torch.zeros_like(utt))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    out[dt] = (sr.double(), g0, flow, utt, ms)
    ploop = ploop.float()
a, b = out[torch.float32], out[torch.float64]
d = (a[0] - b[0]).abs()
print(f'frames {T} spread {spread}: gamma max abs err {float(d.max()):.3e}  mean abs {float(d.mean()):.3e}  '
      f'row sums off by {float((a[0].sum(1) - 1).abs().max()):.3e}')
print(f'  col-sum (state occupancy) max rel err {float(((a[0].sum(0) - b[0].sum(0)).abs() / b[0].sum(0)).max()):.3e}')
print(f'  gamma0 rel {float(((a[1] - b[1]).abs().max() / b[1].abs().max())):.3e}  flow rel '
      f'{float(((a[2] - b[2]).abs().max() / b[2].abs().max())):.3e}  utt_llh rel '
      f'{float(((a[3] - b[3]).abs() / b[3].abs()).max()):.3e}')
print(f'  time f32 {a[4]:.3f} ms  f64 {b[4]:.3f} ms  ({T / a[4] / 1e3:.1f} M frames/s f32)')
