#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=gpurun_out/j10; mkdir -p $O
python - <<'PY' > $O/host.log 2>&1
import sys, time, cProfile, pstats, io
sys.argv=['bench_hmm.py','--cov','diagonal','--steps','5']
sys.path.insert(0,'tools')
import torch, numpy as np
import beer_amd as beer
import bench_hmm as B
dev=torch.device('cuda',0)
rng=np.random.RandomState(2)
lengths=[]
while sum(lengths)<1_000_000: lengths.append(int(rng.randint(200,401)))
total=sum(lengths)
g=torch.Generator(device=dev).manual_seed(2)
X=torch.randn(total,40,generator=g,device=dev)
ploop,units=B.build(40,16,40,'diagonal',dev,torch.float32)
optim=beer.VBConjugateOptimizer(ploop.mean_field_factorization(),1.)
def run():
    optim.init_step()
    elbo=beer.accumulate_elbo(ploop,(X,lengths),datasize=total)
    elbo.backward(); optim.step()
    return elbo
for _ in range(3): run()
import gc; gc.collect(); gc.freeze()
torch.cuda.synchronize()
hs=[]
t0=time.perf_counter()
for _ in range(8):
    a=time.perf_counter(); run(); hs.append(time.perf_counter()-a)
torch.cuda.synchronize()
print('wall per iter ms', 1e3*(time.perf_counter()-t0)/8, 'host per iter ms', [round(1e3*h,2) for h in hs])
pr=cProfile.Profile(); pr.enable()
for _ in range(5): run()
pr.disable(); torch.cuda.synchronize()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats('cumtime').print_stats(45); print(s.getvalue())
PY
head -90 $O/host.log
