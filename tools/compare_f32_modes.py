#!/usr/bin/env python
"""Accumulated statistics of the exact-fp32 and the fp16-split matrix paths against
an fp64 product on the same random inputs (K = 256, D = 40, full covariance).

    python tools/compare_f32_modes.py
"""
import sys, torch, numpy as np
sys.path.insert(0, '/root/repo')
import beer_amd as beer
from beer_amd import kernels
from beer_amd.stats import FrameStats
torch.manual_seed(0)
for T in (8192, 100000, 1000000):
    K, D = 256, 40
    X = torch.randn(T, D, device='cuda') * 3 + 1
    R = torch.softmax(torch.randn(T, K, device='cuda') * 3, dim=1)
    st = FrameStats(X, 'full')
    beer.set_f32_mode('exact')
    a = kernels.normal_accumulate(st, R, None, K, 1, 'full')
    beer.set_f32_mode('bf16x3')
    b = kernels.normal_accumulate(st, R, None, K, 1, 'full')
    ref = (R.double().t() @ st.dense().double())
    ea = ((a - ref).abs().max() / ref.abs().max()).item()
    eb = ((b - ref).abs().max() / ref.abs().max()).item()
    # per-entry relative
    print(T, 'exact err', ea, 'split err', eb, 'count sums', (-2*a[:, -2].sum()).item(), (-2*b[:, -2].sum()).item())
    rel = ((b - ref).abs() / (ref.abs() + 1e-3 * ref.abs().max())).max().item()
    print('   max elementwise rel (split)', rel, ' exact', ((a - ref).abs() / (ref.abs() + 1e-3 * ref.abs().max())).max().item())
