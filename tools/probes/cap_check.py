import sys, os
sys.path.insert(0, '/root/repo')
import torch, bench, beer_amd as beer
dev = torch.device('cuda:0')
lengths = bench.hmm_corpus(300_000)
T = sum(lengths); N = 10 * T
g = torch.Generator(device=dev).manual_seed(2)
X = torch.randn(T, bench.D, generator=g, device=dev)
def run(captured):
    ploop = bench.make_phone_loop('diagonal', dev)
    optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.)
    vals = []
    if captured:
        it = beer.CapturedIteration(ploop, optim, (X, lengths), datasize=N)
        for _ in range(8):
            vals.append((float(it()), it.mode))
    else:
        for _ in range(8):
            optim.init_step()
            elbo = beer.accumulate_elbo(ploop, (X, lengths), datasize=N)
            elbo.backward(); optim.step()
            vals.append((float(elbo), 'eager'))
    w = ploop.categorical.weights.posterior.params.concentrations.cpu()
    return vals, w
a, wa = run(False); b, wb = run(True)
for (x, _), (y, m) in zip(a, b):
    print(f'{x:.10e} {y:.10e} {m} rel {abs(x-y)/abs(x):.2e}')
print('weights diff', float((wa - wb).abs().max()), float(wa.abs().max()))
