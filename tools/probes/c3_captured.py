"""Config 3 (diagonal) iteration eager vs CapturedIteration on a shard:
    python tools/probes/c3_captured.py [frames] [shard_of]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench, beer_amd as beer
from beer_amd.distributed import shard_utterances

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
shard_of = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device('cuda:0')
lengths_all = bench.hmm_corpus(frames)
mine = shard_utterances(lengths_all, shard_of, 0)
lengths = [lengths_all[u] for u in mine]
T = sum(lengths)
g = torch.Generator(device=dev).manual_seed(2)
X = torch.randn(T, bench.D, generator=g, device=dev)
N = sum(lengths_all)


def timed(fn, n=20, warm=4):
    for _ in range(warm):
        v = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        v = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3, float(v)


ploop = bench.make_phone_loop('diagonal', dev)
optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.)
images, statics = beer.FrameImages(X), beer.ShardStatics()


def eager():
    optim.init_step()
    elbo = beer.accumulate_elbo(ploop, (X, lengths), datasize=N, frame_images=images, statics=statics)
    elbo.backward()
    optim.step()
    return elbo.value
ms_e, v_e = timed(eager)
ploop2 = bench.make_phone_loop('diagonal', dev)
optim2 = beer.VBConjugateOptimizer(ploop2.mean_field_factorization(), 1.)
it = beer.CapturedIteration(ploop2, optim2, (X, lengths), datasize=N)
ms_c, v_c = timed(it)
print(f'{T} frames in {len(lengths)} utterances: eager {ms_e:.3f} ms/iteration, captured ({it.mode}) {ms_c:.3f} ms; '
      f'last ELBO {v_e:.6e} / {v_c:.6e} rel diff {abs(v_e - v_c) / abs(v_e):.2e}')
