"""Config 4, diagonal prior: accumulation of the state posteriors over 1 M latent samples --
the default route (`accd_kernel` since round 5; the exact float32 kernel before) against packing the
posteriors first and the packed bf16x3 kernel; then the default route against fp64 for several shapes."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import beer_amd as beer
from beer_amd import kernels
dev = torch.device('cuda:0')
T, D, S = 1_000_037, 64, 120
torch.manual_seed(0)
Z = torch.randn(T, D, device=dev)
W = torch.softmax(torch.randn(T, S, device=dev) * 3, dim=1)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for cov in ('diagonal', 'isotropic'):
    st = beer.FrameStats(Z, cov)
    a = kernels.normal_accumulate(st, W, None, S, 1, cov)
    b = kernels.normal_accumulate(st, kernels.pack_resps(st, W, None, S, 1), None, S, 1, cov)
    ref = torch.cat([W.double().t() @ Z.double(), -.5 * (W.double().t() @ (Z.double() ** 2))], 1) if cov == 'diagonal' else None
    print(cov, 'default route vs packed route: max rel', float(((a - b).abs() / (a.abs() + 1e-3)).max()))
    if ref is not None:
        print('  vs fp64: default', float(((a[:, :2 * D] - ref).abs() / (ref.abs() + 1e-3)).max()),
              'packed', float(((b[:, :2 * D] - ref).abs() / (ref.abs() + 1e-3)).max()))
    print('  default %.3f ms' % t(lambda: kernels.normal_accumulate(st, W, None, S, 1, cov)))
    print('  pack    %.3f ms' % t(lambda: kernels.pack_resps(st, W, None, S, 1)))
    p = kernels.pack_resps(st, W, None, S, 1)
    print('  packed  %.3f ms' % t(lambda: kernels.normal_accumulate(st, p, None, S, 1, cov)))

# parity of the default route against fp64 for several shapes
import numpy as np
for cov, D, K, T in (('diagonal', 64, 120, 100_003), ('diagonal', 40, 256, 70_000), ('isotropic', 24, 48, 33_333),
                     ('diagonal', 7, 300, 20_000), ('isotropic', 64, 17, 16_384), ('diagonal', 33, 64, 50_001)):
    Z = torch.randn(T, D, device=dev) * 1.3 + .2
    W = torch.softmax(torch.randn(T, K, device=dev) * 3, dim=1)
    st = beer.FrameStats(Z, cov)
    a = kernels.normal_accumulate(st, W, None, K, 1, cov)
    Zd, Wd = Z.double(), W.double()
    N = Wd.sum(0)
    if cov == 'diagonal':
        ref = torch.cat([Wd.t() @ Zd, -.5 * (Wd.t() @ Zd ** 2), -.5 * N[:, None], .5 * N[:, None]], 1)
    else:
        ref = torch.cat([Wd.t() @ Zd, -.5 * (Wd.t() @ (Zd ** 2).sum(1, keepdim=True)), -.5 * N[:, None], .5 * D * N[:, None]], 1)
    scale = (Wd.t() @ Zd.abs()).max()
    print(cov, D, K, T, 'max abs err / scale', float((a - ref).abs().max() / scale),
          'counts rel', float(((a[:, -2] * -2 - N).abs() / N).max()))
