#!/usr/bin/env python
"""Time `beer_normal_accumulate` alone at the config-3 shape (K = 1920 = 120 states x
16 components, state responsibilities multiplied in), diagonal and full covariance.

    python tools/time_accumulate.py
"""
import sys, torch, time
sys.path.insert(0, '/root/repo')
import beer_amd as beer
from beer_amd import kernels
from beer_amd.stats import FrameStats
T, S, G, D = 500000, 120, 16, 40
K = S * G
X = torch.randn(T, D, device='cuda')
R = torch.rand(T, K, device='cuda')
SR = torch.rand(T, S, device='cuda')
for cov in ('diagonal', 'full'):
    st = FrameStats(X, cov)
    for _ in range(2): kernels.normal_accumulate(st, R, SR, S, G, cov)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): kernels.normal_accumulate(st, R, SR, S, G, cov)
    torch.cuda.synchronize()
    print(cov, 'ms per call', (time.perf_counter() - t0) / 5 * 1e3)
