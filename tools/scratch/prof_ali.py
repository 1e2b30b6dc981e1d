import sys, time, torch, numpy as np, cProfile, pstats
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import beer_amd as beer
from beer_amd import hmm_kernels as hk
from bench_hmm import build
dev = torch.device('cuda', 0)
rng = np.random.RandomState(2)
ploop, units = build(40, 16, 40, 'diagonal', dev, torch.float32)
lengths = [int(rng.randint(200, 401)) for _ in range(3000)]
seqs = [list(rng.randint(0, 40, max(2, T // 30))) for T in lengths]
gset = beer.graph.compile_alignments(seqs, units)
graphs = list(gset)
gset.device_image(torch.float32)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b = hk.HmmBatch(graphs, list(range(3000)), lengths, torch.float32)
    torch.cuda.synchronize(); print('HmmBatch ms', (time.perf_counter() - t0) * 1e3)
pr = cProfile.Profile(); pr.enable(); b = hk.HmmBatch(graphs, list(range(3000)), lengths, torch.float32); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(8)
from beer_amd.inference import batch as B
from beer_amd import kernels
X = torch.randn(sum(lengths), 40, device=dev)
optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.)
log = []
def wrap(obj, name, tag=None):
    fn = getattr(obj, name)
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(*a, **k); torch.cuda.synchronize(); log.append((tag or name, (time.perf_counter() - t0) * 1e3)); return r
    setattr(obj, name, w)
for m, n in ((hk, 'HmmBatch'), (B, '_emission_estep'), (hk, 'gather'), (hk, 'forward_backward'), (hk, 'scatter'), (kernels, 'normal_accumulate')):
    wrap(m, n)
def run():
    optim.init_step()
    elbo = beer.accumulate_elbo(ploop, (X, lengths), datasize=sum(lengths), inference_graphs=graphs)
    elbo.backward(); optim.step(); torch.cuda.synchronize()
for _ in range(3): run()
log.clear(); t0 = time.perf_counter(); run(); print('ali iteration ms', (time.perf_counter() - t0) * 1e3, log)
import beer_amd.hmm_kernels as hk2
orig_init = hk2.HmmBatch.__init__
calls = [0]
def prof_init(self, *a, **k):
    calls[0] += 1
    if calls[0] % 2 == 0:
        pr = cProfile.Profile(); pr.enable(); orig_init.__wrapped__(self, *a, **k) if hasattr(orig_init, '__wrapped__') else orig_init(self, *a, **k); pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(6)
    else:
        orig_init(self, *a, **k)
hk2.HmmBatch.__init__ = prof_init
B.hk.HmmBatch = hk2.HmmBatch
run()
