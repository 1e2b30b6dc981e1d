"""HIP-event time of beer_frames_llh_backward (csrc/sample_grad.hip) at config 4's shape:
python tools/probes/sgrad_time.py [T] [D] [K]   (BEER_HIP_LIB selects an A/B build)"""
import sys
import torch
sys.path.insert(0, '.')
import beer_amd as beer
from beer_amd import kernels

T = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 64
K = int(sys.argv[3]) if len(sys.argv) > 3 else 120
torch.manual_seed(0)
X = torch.randn(T, D, device='cuda')
E = torch.randn(K, D * D + D + 2, device='cuda') / D ** .5
w = torch.rand(T, K, device='cuda')
g = torch.rand(T, device='cuda')
st = kernels.sample_stats(X, 'full')
for _ in range(3):
    kernels.frames_llh_backward(st, w, g, E)
torch.cuda.synchronize()
n = 10
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(n):
    out = kernels.frames_llh_backward(st, w, g, E)
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / n
print(f'T={T} D={D} K={K}: {ms:.3f} ms  {2. * T * K * D * (D + 1) / ms / 1e9:.0f} TFLOP/s algorithmic '
      f'({6 * 2. * T * K * D * D / ms / 1e9:.0f} executed)  checksum {float(out.double().sum()):.6e}')
