"""How often does the scaled-probability forward-backward hand an utterance to its log-space twin?
Config-3 phone loop (40 x 3 states x 16 Gaussians, D = 40), three kinds of data:
 white:   N(0, I) frames (the bench's)
 model:   frames drawn around the model's own component means (std 1.0), phones of ~30 frames
 sharp:   the same after 3 VB iterations on that data, evaluated with acoustic scale 1 and 5"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, beer_amd as beer
from beer_amd import hmm_kernels as hk
dev = torch.device('cuda:0')
for cov in ('diagonal', 'full'):
    ploop = bench.make_phone_loop(cov, dev)
    lens = bench.hmm_corpus(300_000)
    T = sum(lens)
    rng = np.random.RandomState(1)
    ns = ploop.modelset.original_modelset.modelsets[0].modelset
    mu = ns.means_precisions.posterior.params.mean.cpu().numpy()
    seq = np.repeat(rng.randint(0, 40, T // 30 + 1), 30)[:T]
    comp = 16 * (3 * seq + rng.randint(0, 3, T)) + rng.randint(0, 16, T)
    data = {'white': torch.randn(T, 40, device=dev),
            'model': torch.from_numpy((mu[comp] + rng.randn(T, 40)).astype(np.float32)).to(dev)}
    for name, X in data.items():
        with hk.counting_log_space() as c:
            beer.accumulate_elbo(ploop, (X, lens), datasize=T)
        print(f'{cov:9s} {name:6s}: {int(c.count)} of {len(lens)} utterances in log space')
    optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.)
    X = data['model']
    for it in range(4):
        optim.init_step()
        elbo = beer.accumulate_elbo(ploop, (X, lens), datasize=T)
        elbo.backward(); optim.step()
    for scale in (1., 5., 20.):
        with hk.counting_log_space() as c:
            beer.accumulate_elbo(ploop, (X, lens), datasize=T, scale=scale)
        print(f'{cov:9s} after 4 iterations, acoustic scale {scale:4.0f}: {int(c.count)} of {len(lens)} utterances in log space')
