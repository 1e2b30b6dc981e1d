import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tools')
import beer_amd as beer
from beer_amd import hmm_kernels as hk
from bench_hmm import build
dev = torch.device('cuda', 0)
rng = np.random.RandomState(2)
lengths = []
while sum(lengths) < 1000000: lengths.append(int(rng.randint(200, 401)))
X = torch.randn(sum(lengths), 40, device=dev)
ploop, units = build(40, 16, 40, 'diagonal', dev, torch.float32)
optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.)
T = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); T[name] = T.get(name, 0) + time.perf_counter() - t0
        return r
    return w
hk.HmmBatch.__init__ = timed('HmmBatch', hk.HmmBatch.__init__)
from beer_amd.models import sequence
sequence.PhoneLoop.phone_counts = timed('phone_counts', sequence.PhoneLoop.phone_counts)
sequence.PhoneLoop._on_weights_update = timed('weights_update', sequence.PhoneLoop._on_weights_update)
from beer_amd.inference import batch as B
B._emission_estep = timed('emission_estep', B._emission_estep)
hk.forward_backward = timed('fb', hk.forward_backward); B.hk.forward_backward = hk.forward_backward
optim.step = timed('optim.step', optim.step)
B.hk.gather = timed('gather', hk.gather); B.hk.scatter = timed('scatter', hk.scatter)
from beer_amd import kernels
B.kernels.normal_accumulate = timed('accumulate', kernels.normal_accumulate)
B.kernels.weights_from_acc = timed('weights_from_acc', kernels.weights_from_acc)
B._sub_batches = timed('_sub_batches', B._sub_batches)
B.pack_utterances = timed('pack', B.pack_utterances)
B._finish = timed('_finish', B._finish)
B._groups = timed('_groups', B._groups)
type(ploop).kl_div_posterior_prior = timed('kl', type(ploop).kl_div_posterior_prior)
type(ploop.categorical).accumulate = timed('cat.accumulate', type(ploop.categorical).accumulate)
type(ploop.categorical).sufficient_statistics = timed('cat.suffstats', type(ploop.categorical).sufficient_statistics)
B._hmm_batch = timed('_hmm_batch_total', B._hmm_batch)
from beer_amd import graph
graph.DeviceGraph.__init__ = timed('DeviceGraph', graph.DeviceGraph.__init__)
def run():
    optim.init_step()
    elbo = beer.accumulate_elbo(ploop, (X, lengths), datasize=sum(lengths))
    tb=time.perf_counter(); torch.cuda.synchronize(); elbo.backward(); torch.cuda.synchronize(); T['backward']=T.get('backward',0)+time.perf_counter()-tb; optim.step()
    torch.cuda.synchronize()
run(); run(); T.clear()
t0 = time.perf_counter(); run(); run(); tot = (time.perf_counter() - t0) / 2
print('iteration (serialised)', tot * 1e3, {k: round(v * 500, 2) for k, v in T.items()})
