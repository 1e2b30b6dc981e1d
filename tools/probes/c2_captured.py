"""Config 2 (K = 256 full covariance, D = 40, 1 M frames): the reference-style loop (captured M-step)
against the whole iteration as one HIP graph (`beer.CapturedIteration`)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench, beer_amd as beer
dev = torch.device('cuda:0')
frames = 1 << 20
X = bench.synth_frames(frames, dev, seed=1)
lengths = [8192] * (frames // 8192)
def t(fn, n=20, w=6):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
model = bench.make_gmm(dev)
optim = beer.VBConjugateOptimizer(model.mean_field_factorization(), 1.)
statics = beer.ShardStatics()
def step():
    optim.init_step()
    e = beer.accumulate_elbo(model, (X, lengths), datasize=frames, statics=statics)
    e.backward(); optim.step()
ms = t(step)
print(f'loop      : {ms:.3f} ms/step = {frames / ms / 1e3:.1f} M frames/s')
model = bench.make_gmm(dev)
it = beer.CapturedIteration(model, beer.VBConjugateOptimizer(model.mean_field_factorization(), 1.), (X, lengths), datasize=frames)
ms = t(lambda: it())
print(f'one graph : {ms:.3f} ms/step = {frames / ms / 1e3:.1f} M frames/s ({it.mode})')
