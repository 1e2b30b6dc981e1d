"""Where does the float32 error of an E-step come from?  (GPU; diagnostic, not a test)
Bench shape: K = 256 full, D = 40, T = 65536, random-initialised model (soft posteriors).
Prints, for the exact fp32 MFMA and the bf16x3 kernels, against the fp64 kernels:
per-frame log-normaliser error, responsibility bias per component, statistics error by
block (counts / first / second moments), and the accumulation alone on given responsibilities."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import beer_amd as beer
from beer_amd import kernels, _hip

DEV = 'cuda'
def run(cov, K, D, T, seed=3):
    rng = np.random.RandomState(seed)
    means = rng.randn(K, D) * 2
    A = rng.randn(D, D) * .2 + np.eye(D)
    Xn = (means[rng.randint(0, K, T)] + rng.randn(T, D) @ A).astype(np.float32)
    X = torch.from_numpy(Xn).to(DEV)
    torch.manual_seed(7)
    c0 = torch.from_numpy(np.cov(Xn.T)).float()
    ns = beer.NormalSet.create(X.mean(0).cpu(), c0 if cov == 'full' else c0.diag(), size=K,
                               prior_strength=1., noise_std=1., cov_type=cov)
    model = beer.Mixture.create(ns, prior_strength=1.).to(DEV)
    E32 = ns.means_precisions.natural_form()
    lw32 = model._log_weights().view(1, K)
    E64, lw64 = E32.double(), lw32.double()
    st64, st32 = beer.FrameStats(X.double(), cov), beer.FrameStats(X, cov)
    ln64, r64 = kernels.mixtureset_estep(st64, E64, lw64, 1, K, cov)
    acc64 = kernels.normal_accumulate(st64, r64, None, 1, K, cov)
    N = -2 * acc64[:, -2]
    print(f'== {cov} K={K} D={D} T={T}: N_k min {float(N.min()):.1f} max {float(N.max()):.1f}')
    def blocks(acc):
        e = (acc - acc64).abs()
        sc = acc64.abs().max()
        return dict(all=float(e.max() / sc), first=float(e[:, :D].max() / acc64[:, :D].abs().max()),
                    second=float(e[:, D:-2].max() / acc64[:, D:-2].abs().max()),
                    counts=float((e[:, -2] / acc64[:, -2].abs()).max()),
                    counts_bias=float(((acc[:, -2] - acc64[:, -2]) / acc64[:, -2]).mean()))
    with _hip.exact_f32():
        ln_e, r_e = kernels.mixtureset_estep(st32, E32, lw32, 1, K, cov)
        acc_e = kernels.normal_accumulate(st32, r_e, None, 1, K, cov)
        acc_e64r = kernels.normal_accumulate(st32, r64.float(), None, 1, K, cov)
    ln_f, packed = kernels.mixture_estep_packed(st32, E32, lw32, K, cov)
    r_f = packed.unpack()
    acc_f = kernels.normal_accumulate(st32, packed, None, 1, K, cov)
    acc_f64r = kernels.normal_accumulate(st32, kernels.pack_resps(st32, r64.float(), None, 1, K), None, 1, K, cov)
    for nm, ln, r, acc, accr in (('exact', ln_e, r_e, acc_e, acc_e64r), ('bf16x3', ln_f, r_f, acc_f, acc_f64r)):
        dl = (ln.double() - ln64)
        dr = (r.double() - r64)
        print(f'  {nm:7s} ln: max|err| {float(dl.abs().max()):.2e} mean err {float(dl.mean()):+.2e} | '
              f'r: max|err| {float(dr.abs().max()):.2e}, per-component sum err / N_k: max {float((dr.sum(0).abs() / N).max()):.2e}')
        print(f'          stats  {blocks(acc)}')
        print(f'          stats with the fp64 responsibilities given: {blocks(accr)}')
run('full', 256, 40, 65536)
run('diagonal', 256, 40, 65536)
