import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from beer_amd import kernels
DEV = 'cuda'
def timeit(f, n=5):
    f(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3
for T, Q, S, G in ((65536, 4162, 120, 1), (65536, 4162, 40, 4), (65536, 1642, 120, 4)):
    K = S * G
    st = torch.randn(T, Q, device=DEV); E = torch.randn(K, Q, device=DEV) / Q ** .5
    w = torch.rand(T, K, device=DEV); sr = torch.rand(T, S, device=DEV); g = torch.rand(T, device=DEV)
    fl = 2. * T * Q * K / 1e9
    a = timeit(lambda: kernels.dense_llh(st, E, 30))
    b = timeit(lambda: kernels._llh_backward(w, g, E))
    c = timeit(lambda: kernels.dense_accumulate(st, w, sr, S, G))
    c2 = timeit(lambda: kernels.dense_accumulate(st, w, None, K, 1))
    d = timeit(lambda: torch.matmul(st, E.t()))
    print(f'T={T} Q={Q} K={K}: llh {a:.2f} ms ({fl/a:.0f} TF), backward {b:.2f} ms ({fl/b:.0f} TF), '
          f'accumulate {c:.2f} ms ({fl/c:.0f} TF), without state posteriors {c2:.2f} ms; torch.matmul llh {d:.2f} ms ({fl/d:.0f} TF); '
          f'stats bytes {T*Q*4/1e6:.0f} MB = {T*Q*4/1e6/a:.0f} GB/s in llh')
