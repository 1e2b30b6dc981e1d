import sys, time, torch, numpy as np, gc
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import beer_amd as beer
from beer_amd import hmm_kernels as hk, kernels
from beer_amd.inference import batch as B
from bench_hmm import build
dev = torch.device('cuda', 0)
rng = np.random.RandomState(2)
lengths = []
while sum(lengths) < 1000000: lengths.append(int(rng.randint(200, 401)))
X = torch.randn(sum(lengths), 40, device=dev)
ploop, units = build(40, 16, 40, 'diagonal', dev, torch.float32)
optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.)
log = []
def wrap(mod, name):
    fn = getattr(mod, name)
    def w(*a, **k):
        t0 = time.perf_counter(); r = fn(*a, **k); log.append((name, t0, time.perf_counter())); return r
    setattr(mod, name, w)
wrap(type(ploop), 'phone_counts'); wrap(B, '_like'); wrap(kernels, 'weights_from_acc'); wrap(B, '_normalset'); wrap(B, '_finish'); wrap(hk, 'HmmBatch'); wrap(B, '_emission_estep'); wrap(hk, 'gather'); wrap(hk, 'forward_backward'); wrap(hk, 'scatter'); wrap(kernels, 'normal_accumulate')
def run():
    optim.init_step()
    ta = time.perf_counter()
    elbo = beer.accumulate_elbo(ploop, (X, lengths), datasize=sum(lengths))
    tb = time.perf_counter()
    elbo.backward(); tc = time.perf_counter(); optim.step(); td = time.perf_counter()
    log.append(('ACC', ta, tb)); log.append(('BACKW', tb, tc)); log.append(('STEP%d' % (optim.update_count % 3), tc, td))
    return elbo
for _ in range(3): run()
torch.cuda.synchronize()
gc.collect(); gc.freeze(); print('gc frozen', gc.get_freeze_count())
for i in range(8):
    log.clear(); t0 = time.perf_counter(); run(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('iter', i, 'host %.1f ms, +sync %.1f' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), ' '.join('%s@%.1f+%.1f' % (n[:6], (a - t0) * 1e3, (b - a) * 1e3) for n, a, b in log))
print(torch.cuda.memory_stats()['num_alloc_retries'], torch.cuda.memory_stats()['num_device_alloc'], torch.cuda.memory_stats()['num_device_free'], torch.cuda.memory_reserved() / 2**30)
