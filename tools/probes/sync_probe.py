"""Which host calls of a config-3 VB iteration synchronise with the device?
torch's sync debug mode turns every implicit synchronisation (.item(), pageable copies,
nonzero ...) into a warning; the probe prints where they come from."""
import sys, os, warnings, traceback, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench, beer_amd as beer
from beer_amd.distributed import shard_utterances, all_reduce_elbo

dev = torch.device('cuda:0')
cov = sys.argv[1] if len(sys.argv) > 1 else 'diagonal'
lengths = bench.hmm_corpus(int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000)
X = torch.randn(sum(lengths), bench.D, device=dev)
ploop = bench.make_phone_loop(cov, dev)
optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.)
def step():
    optim.init_step()
    elbo = beer.accumulate_elbo(ploop, (X, lengths), datasize=sum(lengths))
    elbo, _ = all_reduce_elbo(elbo, ploop, len(lengths))
    elbo.backward()
    optim.step()
    return elbo
for _ in range(2):
    step()
torch.cuda.synchronize()
seen = collections.Counter()
def show(message, category, filename, lineno, file=None, line=None):
    stack = [f for f in traceback.extract_stack() if '/beer_amd/' in f.filename or 'bench.py' in f.filename]
    key = ' <- '.join(f'{os.path.basename(f.filename)}:{f.lineno}' for f in reversed(stack[-4:]))
    seen[(str(message)[:60], key)] += 1
warnings.showwarning = show
torch.cuda.set_sync_debug_mode(1)
step(); step()
torch.cuda.set_sync_debug_mode(0)
torch.cuda.synchronize()
for (m, k), n in seen.most_common():
    print(n, m, '|', k)
print('done', len(seen))
