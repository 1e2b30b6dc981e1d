import sys, torch, os
sys.path.insert(0, '/root/repo')
import beer_amd as beer
from beer_amd import kernels
from beer_amd.stats import FrameStats
T, K, D = 1000000, 256, 40
X = torch.randn(T, D, device='cuda')
R = torch.softmax(torch.randn(T, K, device='cuda') * 3, dim=1)
st = FrameStats(X, 'full')
for _ in range(2): kernels.normal_accumulate(st, R, None, K, 1, 'full')
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5): kernels.normal_accumulate(st, R, None, K, 1, 'full')
b.record(); torch.cuda.synchronize()
print('BEER_DBG', os.environ.get('BEER_DBG'), 'ms', a.elapsed_time(b) / 5)
