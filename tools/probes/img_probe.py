import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import beer_amd as beer
from beer_amd import kernels, _hip
DEV = 'cuda'
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3333335
D, S, G = 40, 120, 16
torch.manual_seed(1)
X = torch.randn(T, D, device=DEV)
K = S * G
ns = beer.NormalSet.create(torch.zeros(D), torch.ones(D), size=K, prior_strength=1., noise_std=1.5, cov_type='diagonal')
E = ns.means_precisions.natural_form().float().to(DEV)
lw = torch.log_softmax(torch.randn(S, G, device=DEV), dim=1)
st = beer.FrameStats(X, 'diagonal')
ln, _ = kernels.mixtureset_estep(st, E, lw, S, G, 'diagonal', want_resps=False)
sr = torch.softmax(torch.randn(T, S, device=DEV), dim=1)
torch.cuda.synchronize(); print('estep done', flush=True)
img = kernels.frame_image(X, 'diagonal')
torch.cuda.synchronize(); print('image done', img.numel(), flush=True)
a = kernels.mixtureset_accumulate_fused(st, E, lw, ln, sr, S, G, 'diagonal')
torch.cuda.synchronize(); print('accfi done', flush=True)
os.environ['BEER_FRAME_IMAGE'] = '0'
b = kernels.mixtureset_accumulate_fused(st, E, lw, ln, sr, S, G, 'diagonal')
torch.cuda.synchronize()
print('max diff', float((a - b).abs().max()) / float(b.abs().max()))
