import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
from beer_amd import _hip
dev = torch.device('cuda', 0)
x = torch.randn(1000000, 40, device=dev)
tens = {'a': torch.arange(3001, dtype=torch.int64), 'b': torch.arange(3000, dtype=torch.int64), 'c': torch.zeros(3000, dtype=torch.int32),
        'g': torch.zeros(3000 * 136, dtype=torch.uint8), 'p': torch.zeros(3001, dtype=torch.int32), 'q': torch.zeros(90000, dtype=torch.int32)}
for i in range(6):
    y = x * 2  # some GPU work
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = _hip.upload(tens, dev)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('upload host %.2f ms, sync %.2f ms' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
# per piece
host, ev = _hip._staging(1 << 20)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter(); blob = torch.empty(500000, dtype=torch.uint8, device=dev); t1 = time.perf_counter()
    blob.copy_(host[:500000], non_blocking=True); t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print('empty %.2f copy_ %.2f sync %.2f' % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
