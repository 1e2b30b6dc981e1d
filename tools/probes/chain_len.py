"""Statistics error of the bf16x3 accumulation alone (fp64 responsibilities given) against the
fp64 kernels, as a function of the MFMA chain length (BEER_OPT_AX_MAXFRAMES, set per sweep
point with _hip.set_option).  Prints one JSON object: {chain: {block errors, biases}}.
    python tools/probes/chain_len.py > profiles/rNN_chain_len.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import beer_amd as beer
from beer_amd import _hip, kernels

DEV = 'cuda'
K, D, T = 256, 40, 1 << 20
rng = np.random.RandomState(3)
means = rng.randn(K, D) * 2
Xn = (means[rng.randint(0, K, T)] + rng.randn(T, D)).astype(np.float32)
X = torch.from_numpy(Xn).to(DEV)
torch.manual_seed(7)
ns = beer.NormalSet.create(X.mean(0).cpu(), torch.diag(X.var(0).cpu()), size=K, prior_strength=1.,
                           noise_std=1., cov_type='full')
model = beer.Mixture.create(ns, prior_strength=1.).to(DEV)
E, lw = ns.means_precisions.natural_form(), model._log_weights().view(1, K)
st64, st32 = beer.FrameStats(X.double(), 'full'), beer.FrameStats(X, 'full')
_, r64 = kernels.mixtureset_estep(st64, E.double(), lw.double(), 1, K, 'full')
acc64 = kernels.normal_accumulate(st64, r64, None, 1, K, 'full')
packed = kernels.pack_resps(st32, r64.float(), None, 1, K)
diag = torch.arange(D, device=DEV) * (D + 1) + D
out = {'shape': {'K': K, 'D': D, 'T': T, 'cov': 'full'},
       'what': 'beer_normal_accumulate_packed on float32(fp64 responsibilities) vs the fp64 kernels',
       'chains': {}}
old = _hip.get_option('ax_max_frames')
for chain in (512, 1024, 2048, 4096, 8192, 16384, 65536):
    _hip.set_option('ax_max_frames', chain)
    acc = kernels.normal_accumulate(st32, packed, None, 1, K, 'full')
    e = (acc - acc64).abs()
    cnt, cnt64 = -2 * acc[:, -2], -2 * acc64[:, -2]
    out['chains'][str(chain)] = {
        'all_max_rel': float(e.max() / acc64.abs().max()),
        'first_moments_max_rel': float(e[:, :D].max() / acc64[:, :D].abs().max()),
        'second_moments_max_rel': float(e[:, D:-2].max() / acc64[:, D:-2].abs().max()),
        'counts_max_rel': float(((cnt - cnt64).abs() / cnt64).max()),
        'counts_bias_mean_rel': float(((cnt - cnt64) / cnt64).mean()),
        'squares_bias_mean_rel': float(((acc[:, diag] - acc64[:, diag]) / acc64[:, diag]).mean())}
_hip.set_option('ax_max_frames', old)
print(json.dumps(out, indent=1))
