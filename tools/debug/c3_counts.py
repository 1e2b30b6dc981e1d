import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
import beer_amd as beer
dev = torch.device('cuda')
lengths = bench.hmm_corpus(200000)
X = torch.randn(sum(lengths), bench.D, device=dev)
ploop = bench.make_phone_loop('diagonal', dev)
elbo = beer.accumulate_elbo(ploop, (X, lengths), datasize=sum(lengths))
for p, v in elbo._acc_stats.items():
    print(tuple(v.shape), v.dtype, float(v.double().sum()), float(v.double()[..., -1].sum()) if v.dim() else None)
print('frames', sum(lengths), 'utts', len(lengths))
