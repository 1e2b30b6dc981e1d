#!/usr/bin/env python
"""Headline benchmark: frames/sec per VB iteration (E-step + all-reduce + M-step).

    python bench.py --gpus N --steps K --warmup W            # the driver's line
    python bench.py --gpus N --config 3 [--cov full] [--frames 10000000]   # config 3 alone

The default line is BASELINE config 2 (`configs[1]`, the configuration the metric is
quoted on): GMM, K = 256 full-covariance Gaussians, D = 40, 1,000,000 fp32 frames PER
GPU in 8192-frame "utterances" (weak scaling; `--scaling strong` shards 1 M frames over
the ranks instead).  It carries two sub-objects for BASELINE config 3 (`configs[2]`),
measured by the same process(es) right after: `config3` -- monophone phone-loop HMM, 40
phones x 3 states x 16 diagonal Gaussians, D = 40, a FIXED corpus of 10 M frames in ~33 k
utterances sharded over the ranks by frame count (strong scaling), forward-backward + VB
update -- and `config3_full`, the same model with full covariances on 2 M frames
(`--no-config3` skips both).

float32 models multiply on the bf16 matrix pipes with every operand held exactly as three
bf16 pieces and six partial products per multiplication (fp32 accumulation): operands and
accumulation are float32's own -- `value` is on that arithmetic; `f32_exact` repeats the
iteration on the exact fp32 MFMA.

One process per GPU, ONE RCCL all-reduce of the accumulated statistics per
iteration, replicated M-step.  With --gpus N > 1 and no launcher in the
environment (no WORLD_SIZE) the script spawns its own N ranks; under
`python -m torch.distributed.run` it takes RANK / LOCAL_RANK / WORLD_SIZE from
the environment.  Rank 0 prints ONE compact JSON line on stdout (benchlib/format.py: the
contract's keys, the headline's roofline and cpu_baseline, a `summary` of the other
configurations; < 4 KB); the full objects -- per-kernel tables, the other configurations' own
rooflines and baselines -- go to bench_detail.json and to stderr.
"""

import argparse
import json
import os
import socket
import sys
import time

# (multi-process GPU work on this pool needs dmabuf IPC; set before the HIP runtime starts -- the
#  driver exports it too, this covers a bare `torchrun bench.py`)
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import numpy as np                                        # noqa: E402
import torch                                              # noqa: E402
import torch.distributed as dist                          # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import beer_amd as beer                                   # noqa: E402
from beer_amd import _hip                                  # noqa: E402
from beer_amd.distributed import all_reduce_elbo, shard_utterances   # noqa: E402

from benchlib.shapes import (D, K, LATENT, N_COMP, N_PHONES, PEAK_HBM_GBS, PEAK_TFLOPS, Q,   # noqa: E402,F401
                             TOPO)
from benchlib.timers import (ClockProbe, KernelTimer, PhaseTimer, kernel_times_entry, pmc_entry,            # noqa: E402,F401
                             pmc_traffic, profiled)
from benchlib.baselines import (cpu_baseline_config1, cpu_baseline_config5,                     # noqa: E402,F401
                                cpu_baseline_features, cpu_baseline_gmm,
                                cpu_baseline_graph_compile, cpu_baseline_hmm,
                                cpu_baseline_vae_prior, gmm_parity_check, host_cores)
from benchlib.format import emit, summary                 # noqa: E402,F401
from benchlib import recipe                               # noqa: E402


# --------------------------------------------------------------------------------------------
# shared pieces
# --------------------------------------------------------------------------------------------


def m_step_mode(optim):
    'How the conjugate M-step ran: replayed from a captured HIP graph, or kernel by kernel.'
    if any(e not in (None, False) for e in getattr(optim, '_captured', {}).values()):
        return 'hipGraph (one capture per mean-field group, replayed every iteration)'
    return 'eager (a parameter of the group has host callbacks, or --no-mstep-graph)' \
        if getattr(optim, 'graph', False) else 'eager'


def rank_census(world, device, backend, n_local):
    '''Evidence for an N > 1 line: the number of ranks the collective really spans (an
    all-reduce of ones) and the frames per rank (max / mean).'''
    if world == 1:
        return {'ranks': 1, 'backend': None,
                'frames_per_rank': {'max': n_local, 'mean': float(n_local)}}
    dev = device if backend == 'nccl' else 'cpu'
    t = torch.tensor([1., float(n_local)], dtype=torch.float64, device=dev)
    m = torch.tensor([float(n_local)], dtype=torch.float64, device=dev)
    dist.all_reduce(t)
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return {'ranks': int(t[0].item()), 'backend': 'rccl' if backend == 'nccl' else backend,
            'frames_per_rank': {'max': int(m.item()), 'mean': float(t[1].item()) / world}}


def fence(world):
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(elapsed, world, device, backend):
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    return elapsed


# --------------------------------------------------------------------------------------------
# config 2: GMM K = 256 full covariance
# --------------------------------------------------------------------------------------------

def synth_frames(n, device, seed):
    'Seeded draw from a 256-component ground-truth mixture in 40 dimensions.'
    g = torch.Generator(device='cpu').manual_seed(seed)
    means = torch.randn(K, D, generator=g) * 2.
    # random SPD covariances with eigenvalues in [0.5, 2]
    A = torch.linalg.qr(torch.randn(K, D, D, generator=g))[0]
    ev = torch.rand(K, D, generator=g) * 1.5 + .5
    chol = (A * ev.sqrt()[:, None, :]).to(device)
    means = means.to(device)
    gd = torch.Generator(device=device).manual_seed(seed + 1)
    X = torch.empty(n, D, dtype=torch.float32, device=device)
    per = (n + K - 1) // K
    for k in range(K):
        lo, hi = k * per, min(n, (k + 1) * per)
        if lo >= hi:
            break
        eps = torch.randn(hi - lo, D, generator=gd, device=device)
        X[lo:hi] = means[k] + eps @ chol[k].t()
    perm = torch.randperm(n, generator=gd, device=device)
    return X[perm].contiguous()


def make_gmm(device):
    '''Mixture of K full-covariance Gaussians initialised from a common seeded
    sample (identical on every rank); init noise drawn once on the CPU.'''
    torch.manual_seed(7)
    X = synth_frames(1 << 17, device, seed=12345)
    mean = X.mean(0).cpu()
    cov = torch.cov(X.t()).cpu()
    ns = beer.NormalSet.create(mean, cov, size=K, prior_strength=1., noise_std=1.,
                               cov_type='full')
    return beer.Mixture.create(ns, prior_strength=1.).to(device)


def run_gmm(args, rank, world, device, backend):
    frames = args.frames
    if args.scaling == 'strong':              # a fixed corpus of args.frames, sharded
        frames = args.frames // world + (1 if rank < args.frames % world else 0)
    X = synth_frames(frames, device, seed=1 + rank)
    lengths = [args.chunk] * (frames // args.chunk)
    if frames % args.chunk:
        lengths.append(frames % args.chunk)
    datasize = args.frames * world if args.scaling == 'weak' else args.frames
    census = rank_census(world, device, backend, frames)
    model = make_gmm(device)             # identical on every rank
    # (the captured M-step at any world size: the capture is thread-local -- a collective's
    # watchdog thread keeps calling into the runtime -- and a refused capture falls back to
    # the eager update, optimizers.py)
    optim = beer.VBConjugateOptimizer(model.mean_field_factorization(), lrate=1.,
                                      graph=False if args.no_mstep_graph else None)
    parity = {}
    if rank == 0 and not args.no_check:
        parity = gmm_parity_check(model, X, n=min(65536, frames))
    mode_now = beer.get_f32_mode()
    elbo_err = parity.get(mode_now, {}).get('elbo_rel_err')
    stats_err = parity.get(mode_now, {}).get('stats_rel_err')
    phases = PhaseTimer()

    statics = beer.ShardStatics()        # offsets / weights of the shard's utterances: the caller's

    def step():
        optim.init_step()
        elbo = beer.accumulate_elbo(model, (X, lengths), datasize=datasize, statics=statics)
        with phases.span('all_reduce'):
            elbo, _ = all_reduce_elbo(elbo, model, len(lengths))
        with phases.span('m_step'):
            elbo.backward()
            optim.step()
        return elbo

    # the float32 split path hands the responsibilities over packed (two entry points)
    names = ('beer_mixtureset_estep', 'beer_normal_accumulate',
             'beer_mixture_estep_packed', 'beer_normal_accumulate_packed')

    def timed_loop(steps, warmup):
        for _ in range(warmup):
            step()
        phases.clear()
        fence(world)
        with KernelTimer(names) as kt:
            t0 = time.perf_counter()
            for _ in range(steps):
                elbo = step()
            fence(world)
            elapsed = time.perf_counter() - t0
        return max_over_ranks(elapsed, world, device, backend), kt, elbo

    def kernel_table(kt, steps):
        kern = {}
        for nm in names:
            ms, n = kt.mean_ms(nm)
            if n == 0:
                continue
            # algorithmic work of one launch (2*K*Q flop per frame per GEMM,
            # no symmetry discount), for the frames one launch processes
            flops = 2. * K * Q * frames * steps / max(1, n)
            kern[nm] = {'ms': ms, 'launches': n, 'tflops': flops / (ms * 1e-3) / 1e12}
        return kern

    elapsed, kt, elbo = timed_loop(args.steps, args.warmup)
    kern = kernel_table(kt, args.steps)
    # the shader clock the iteration's kernels run at on THIS box (a sleeping wave on a side stream)
    # (every rank: the step holds a collective)
    clock = None if args.no_extras else ClockProbe(device).measure(step, 1e3 * elapsed / args.steps)
    allreduce_ms, mstep_ms = phases.mean_ms('all_reduce'), phases.mean_ms('m_step')
    mode = beer.get_f32_mode()
    # secondary: the same iteration on the exact fp32 MFMA (every product an fmaf)
    exact = None
    if mode == 'bf16x3' and not args.no_exact:
        with _hip.exact_f32():
            e_steps = max(2, min(5, args.steps))
            e_elapsed, e_kt, _ = timed_loop(e_steps, 1)
            e_kern = kernel_table(e_kt, e_steps)
        e_dom = max(e_kern, key=lambda nm: e_kern[nm]['ms'] * e_kern[nm]['launches'])
        # what the exact kernels execute: the D(D+1)/2 + D + 1 distinct products of a frame's
        # statistics in slabs of 4 (231 slabs = 924 of the Q = 1642 entries), not SURVEY 8d's
        # dense 2*K*Q -- `frac` is priced in those executed multiply-adds, so it cannot exceed 1
        q_exec = 4 * (D // 4 * (D // 4 + 1) // 2 * 4 + D // 4 + 1)
        exact = {'ms_per_step': 1e3 * e_elapsed / e_steps,
                 'value': datasize * e_steps / e_elapsed, 'kernel': e_dom,
                 'achieved': e_kern[e_dom]['tflops'] * q_exec / Q, 'peak': PEAK_TFLOPS['f32'],
                 'frac': e_kern[e_dom]['tflops'] * q_exec / Q / PEAK_TFLOPS['f32'],
                 'achieved_algorithmic': e_kern[e_dom]['tflops'],
                 'avg_launch_ms': e_kern[e_dom]['ms'],
                 'traffic': pmc_traffic('acc_kernel' if 'accumulate' in e_dom else 'llh_kernel'),
                 **{k_: v_ for k_, v_ in parity.get('exact', {}).items()},
                 'note': 'v_mfma_f32_16x16x4_f32, bitwise an fmaf chain.  achieved / frac count '
                         f'the multiply-adds the kernels execute (2*K*{q_exec} per frame: the '
                         'distinct products of the symmetric statistics); achieved_algorithmic is '
                         f'SURVEY 8d\'s 2*K*Q = 2*K*{Q} per frame over the same time'}
    if rank != 0:
        return None
    ms_per_step = 1e3 * elapsed / args.steps
    dom = max(kern, key=lambda nm: kern[nm]['ms'] * kern[nm]['launches'])
    split = mode == 'bf16x3'
    peak = PEAK_TFLOPS['bf16' if split else 'f32']
    if split:
        note = ('every fp32 operand is held exactly as three bf16 pieces and every product is '
                'the six leading partial products on v_mfma_f32_16x16x32_bf16 (fp32 accumulate): '
                'operands and accumulation are fp32, product error <= 2^-23.  The peak is the '
                'dense bf16 MFMA peak; achieved = algorithmic flops (2*K*Q per frame, no symmetry '
                'discount, one flop pair per product) / HIP-event time of the C-ABI call.  The '
                'matrix cores execute 6 * 2*K*928 flop per frame (3.39x the algorithmic count): '
                'hardware rate = 3.39 * achieved.  f32_exact: the same iteration on the exact '
                'fp32 MFMA (peak 157.3).')
    else:
        note = ('achieved = algorithmic flops (2*K*Q per frame, no symmetry discount) / '
                'HIP-event time of the C-ABI call; the kernels contract only the D(D+1)/2 '
                'symmetric products (0.56x the multiply-adds), so frac can exceed 1')
    pmc_key = ('accx_kernel' if split else 'acc_kernel') \
        if 'accumulate' in dom else ('llhx_kernel' if split else 'llh_kernel')
    out = {
        'metric': 'frames/sec per VB iteration (E+M)', 'value': datasize * args.steps / elapsed,
        'unit': 'frames/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': args.scaling,
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': 'configs[1]: GMM K=256 full-covariance, D=40, '
                               + (f'{args.frames} fp32 frames per GPU' if args.scaling == 'weak'
                                  else f'{args.frames} fp32 frames sharded over the GPUs')
                               + f' in {args.chunk}-frame utterances, 1 VB iteration = E-step + '
                               'all-reduce + M-step',
                   'parallelism': f'dp{world}', 'frames_per_gpu': frames,
                   'components': K, 'dim': D},
        **census,
        'elbo_rel_err_vs_cpu_fp64': elbo_err, 'stats_rel_err_vs_cpu_fp64': stats_err,
        'parity_vs_cpu_fp64': parity,
        'parity_check': 'first 65536 frames through the timed kernels (packed hand-over) vs the '
                        'fp64 numpy oracle, in both float32 arithmetics; statistics per block '
                        '(counts / first / second moments against their own largest entry)',
        'elbo_per_frame': float(elbo) / (len(lengths) * world * datasize),
        'f32_mode': mode,
        'f32_arithmetic': 'bf16x3: every fp32 operand exactly as three bf16 pieces, six partial '
                          'products per multiplication on v_mfma_f32_16x16x32_bf16, fp32 '
                          'accumulation (not narrower than fp32: 24-bit operands, product error '
                          '<= 2^-23)' if split else 'v_mfma_f32_16x16x4_f32',
        'all_reduce_ms': allreduce_ms, 'm_step_ms': mstep_ms,
        'm_step': m_step_mode(optim), 'clock': clock,
        'roofline': {'bound': 'mfma', 'kernel': dom, 'achieved': kern[dom]['tflops'],
                     'peak': peak, 'unit': 'TFLOP/s', 'frac': kern[dom]['tflops'] / peak,
                     'avg_launch_ms': kern[dom]['ms'], 'note': note},
        'kernels': kern,
    }
    out['roofline'].update(profiled(pmc_key, out['roofline']))
    if exact:
        out['f32_exact'] = exact
    if world == 1 and not args.no_extras:
        # the same iterations recorded ONCE as a HIP graph and replayed (beer.CapturedIteration:
        # E-step, statistics, KL and the update as one submission): what the host-side launches of
        # the loop above cost.  Not the headline: a replay has no per-call HIP events.
        m2 = make_gmm(device)
        it = beer.CapturedIteration(m2, beer.VBConjugateOptimizer(m2.mean_field_factorization(), 1.),
                                    (X, lengths), datasize=datasize)
        for _ in range(max(4, args.warmup)):
            it()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            it()
        torch.cuda.synchronize()
        c_ms = 1e3 * (time.perf_counter() - t0) / args.steps
        out['captured'] = {'ms_per_step': c_ms, 'value': datasize / (c_ms * 1e-3), 'mode': it.mode,
                           'note': 'beer.CapturedIteration: the whole iteration as one HIP graph'}
        del m2, it
    if not args.no_cpu_baseline and world == 1:
        out['cpu_baseline'] = cpu_baseline_gmm()
    return out


# --------------------------------------------------------------------------------------------
# config 3: monophone phone-loop HMM
# --------------------------------------------------------------------------------------------



def make_phone_loop(cov, device, dim=None, n_comp=None):
    '''beer hmm mkphones / mkphoneloopgraph / mkphoneloop in memory (recipes/aud/conf/hmm.yml
    topology).  `n_comp` = 1: one Gaussian per state (the prior of config 4's HMM-VAE, over
    its `dim`-dimensional latent variable).'''
    dim, n_comp = dim or D, n_comp or N_COMP
    units, pdf = {}, 0
    for p in range(N_PHONES):
        g = beer.graph.Graph()
        for sid in range(5):
            g.add_state(pdf_id=None if sid in (0, 4) else pdf + sid - 1)
        g.start_state, g.end_state = 0, 4
        for arc in TOPO:
            g.add_arc(*arc)
        units[p] = g
        pdf += 3
    graph = beer.graph.Graph()
    graph.start_state, graph.end_state = graph.add_state(), graph.add_state()
    pivot = graph.add_state()
    u2s = {p: graph.add_state() for p in units}
    graph.add_arc(graph.start_state, pivot)
    graph.add_arc(pivot, graph.end_state)
    for p in units:
        graph.add_arc(pivot, u2s[p])
        graph.add_arc(u2s[p], pivot)
    graph.normalize()
    for p, hmm in units.items():
        graph.replace_state(u2s[p], hmm)
    graph.normalize()
    torch.manual_seed(3)
    S = 3 * N_PHONES
    ns = beer.NormalSet.create(torch.zeros(dim), torch.ones(dim), size=S * n_comp, prior_strength=1.,
                               noise_std=1., cov_type=cov)
    emissions = ns if n_comp == 1 else \
        beer.JointModelSet([beer.MixtureSet.create(S, ns, prior_strength=1.)])
    ploop = beer.PhoneLoop.create(graph.compile(), {p: 3 * p for p in units},
                                  {p: 3 * p + 2 for p in units}, emissions)
    return ploop.float().to(device)


def hmm_corpus(total_frames):
    'Utterance lengths U[200, 400] until `total_frames` (seed 2): the same corpus for every N.'
    rng = np.random.RandomState(2)
    lengths = []
    while sum(lengths) < total_frames:
        lengths.append(int(rng.randint(200, 401)))
    return lengths


def run_hmm(args, rank, world, device, backend, cov=None, total_frames=None, steps=None,
            warmup=None, with_cpu_baseline=True, shard_of=None):
    '''`shard_of` = N (one process): this GPU takes the part of the corpus that rank 0 of an
    N-way split would take (`shard_utterances`: balanced by frame count) -- the per-rank work
    of an N-GPU run measured on one GPU; no collective runs.'''
    cov = cov or args.cov
    total_frames = total_frames or args.frames
    steps, warmup = steps or args.steps, args.warmup if warmup is None else warmup
    lengths_all = hmm_corpus(total_frames)
    datasize = sum(lengths_all)
    if shard_of:
        mine = shard_utterances(lengths_all, shard_of, 0)
    else:
        mine = shard_utterances(lengths_all, world, rank)    # balanced by frame count
    lengths = [lengths_all[u] for u in mine]
    n_local = sum(lengths)
    g = torch.Generator(device=device).manual_seed(2 + rank)
    X = torch.randn(n_local, D, generator=g, device=device)
    ploop = make_phone_loop(cov, device)                     # identical on every rank
    census = rank_census(world, device, backend, n_local)
    optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.,
                                      graph=False if args.no_mstep_graph else None)
    phases = PhaseTimer()
    # the frames stay resident over the iterations, and so do their fragment images: the
    # caller (this script) owns both
    images = beer.FrameImages(X) if cov != 'full' and not os.environ.get('BEER_BENCH_NO_IMAGES') \
        else None

    statics = beer.ShardStatics()        # offsets / weights of the shard's utterances: the caller's

    def step():
        optim.init_step()
        elbo = beer.accumulate_elbo(ploop, (X, lengths), datasize=datasize,
                                    max_frames=args.max_frames, frame_images=images,
                                    statics=statics)
        with phases.span('all_reduce'):
            elbo, _ = all_reduce_elbo(elbo, ploop, len(lengths))
        with phases.span('m_step'):
            elbo.backward()
            optim.step()
        return elbo

    names = ('beer_mixtureset_estep', 'beer_mixtureset_lognorm_image', 'beer_hmm_posteriors_fused', 'beer_hmm_forward_backward',
             'beer_mixtureset_accumulate_fused', 'beer_normal_accumulate',
             'beer_normal_accumulate_packed', 'beer_pack_resps',
             'beer_mixtureset_estep_packed', 'beer_mixtureset_accumulate_packed')
    for _ in range(warmup):
        step()
    import gc
    gc.collect()
    gc.freeze()                  # the set-up's objects: a gen-2 collection over them costs 30 ms
    phases.clear()
    fence(world)
    with KernelTimer(() if os.environ.get('BEER_BENCH_NO_KT') else names) as kt:
        t0 = time.perf_counter()
        for _ in range(steps):
            elbo = step()
        fence(world)
        elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, world, device, backend)
    # (every rank: a collective)
    clock = None if args.no_extras else ClockProbe(device).measure(step, 1e3 * elapsed / steps)
    if os.environ.get('BEER_BENCH_NO_KT'):
        print('NO_KT ms/step', elapsed / steps * 1e3, file=sys.stderr)
        return None
    if rank != 0:
        return None
    # size-independent check at the full size: every frame's state posteriors and
    # component responsibilities sum to one, so the mixture-weight counts of the
    # whole (all-reduced, datasize-scaled) job must add up to the number of frames
    counts = [v for p, v in elbo._acc_stats.items() if v.dim() == 2 and v.shape[-1] == N_COMP]
    # (a row of the statistics is (N_1 .. N_G-1, N_1 + .. + N_G): beer/dists/dirichlet.py:18-21)
    expect = n_local if shard_of else datasize
    conservation = abs(float(counts[0].double()[:, -1].sum()) - expect) / expect \
        if counts else None
    Kc = 3 * N_PHONES * N_COMP
    Qd = {'diagonal': 2 * D + 2, 'full': D * D + D + 2, 'isotropic': D + 3}[cov]
    kern = {}
    for nm in names:
        ms, n = kt.mean_ms(nm)
        if n:
            fpl = n_local * steps / n
            kern[nm] = {'ms': ms, 'launches': n, 'frames_per_launch': fpl}
            if 'estep' in nm or 'accumulate' in nm or 'lognorm' in nm:
                # 2*K*Q algorithmic flop per frame for each of the two products
                kern[nm]['tflops'] = 2. * Kc * Qd * fpl / (ms * 1e-3) / 1e12
    dom = max(kern, key=lambda nm: kern[nm]['ms'] * kern[nm]['launches'])
    kname = {'beer_mixtureset_accumulate_fused': 'accf_kernel',
             'beer_mixtureset_estep_packed': 'llhx_kernel',
             'beer_mixtureset_accumulate_packed': 'accx_kernel',
             'beer_mixtureset_estep': 'llhx_kernel',
             'beer_mixtureset_lognorm_image': 'lnfi_kernel',
             'beer_hmm_posteriors_fused': 'fb_wave_kernel'}.get(dom, dom)
    pmc_key = ('c3full_' if cov == 'full' else 'c3_') + kname
    pmc = pmc_entry(pmc_key)
    # SURVEY 8d names the roofline: matrix pipe for full covariances; VALU + transcendental
    # throughput for diagonal ones (Q is small: per frame and Gaussian one exponential, the
    # three-way split of its responsibility and a share of the fragment arithmetic stand
    # beside 4 Q multiply-adds).  Peaks: dense bf16 MFMA (the pipe the kernels run on) /
    # the fp32 vector rate.
    bound = 'mfma' if cov == 'full' else 'valu'
    peak = PEAK_TFLOPS['bf16' if cov == 'full' else 'f32']
    achieved = kern[dom].get('tflops')
    if achieved is None:                     # (the forward-backward call dominating: no flop count)
        bound, peak, achieved = 'hbm', PEAK_HBM_GBS, 4. * D * kern[dom]['frames_per_launch'] / \
            (kern[dom]['ms'] * 1e-3) / 1e9
    fpl = kern[dom]['frames_per_launch']
    roof = {'bound': bound, 'kernel': dom, 'achieved': achieved, 'peak': peak,
            'unit': 'GB/s' if bound == 'hbm' else 'TFLOP/s', 'frac': achieved / peak,
            'avg_launch_ms': kern[dom]['ms'],
            'frac_of_bf16_mfma_peak': None if bound == 'hbm' else achieved / PEAK_TFLOPS['bf16'],
            'valu_wave_insts_per_frame': (pmc['SQ_INSTS_VALU'] / fpl)
            if 'SQ_INSTS_VALU' in pmc and not shard_of else None,
            'counter_bytes_per_frame': (pmc_traffic(pmc_key) / fpl)
            if pmc_traffic(pmc_key) and not shard_of else None,
            'algorithmic_bytes_per_frame': 4 * D,
            'note': 'achieved = algorithmic flops of the dominant call (2*K*Q per frame: one of '
                    'the two products of the iteration; the fused accumulation also recomputes '
                    'the logits, which is not counted) / its HIP-event time.  `valu`: the fp32 '
                    'vector peak (SURVEY 8d: VALU + transcendental bound); the kernels run their '
                    'products on the bf16 MFMA with three pieces per operand (6 MFMAs per '
                    f'product), see frac_of_bf16_mfma_peak.  Iteration: 4*K*Q = {4 * Kc * Qd} '
                    'flop per frame, 160 B of frames per frame.'}
    # (a shard's launches are smaller than the profiled full-size ones: no profiled fraction)
    roof.update(profiled(None if shard_of else pmc_key, roof))
    out = {
        'metric': 'frames/sec per VB iteration (E+M)', 'value': datasize * steps / elapsed,
        'unit': 'frames/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
        'ms_per_step': 1e3 * elapsed / steps, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'configs[2]: monophone phone-loop HMM, {N_PHONES} phones x 3 states, '
                               f'{N_COMP} {cov} Gaussians per state (K={Kc}), '
                               f'D={D}, {datasize} fp32 frames in {len(lengths_all)} utterances '
                               f'sharded over the ranks by frame count (free phone loop), 1 VB '
                               'iteration = emission E-step + forward-backward + statistics + '
                               'all-reduce + M-step',
                   'parallelism': f'dp{world}', 'frames_total': datasize,
                   'frames_rank0': n_local, 'utterances': len(lengths_all)},
        **census,
        'elbo_per_frame': float(elbo) / (len(lengths_all) * datasize),
        'count_conservation_rel_err': conservation,
        'f32_mode': beer.get_f32_mode(),
        'all_reduce_ms': phases.mean_ms('all_reduce'), 'm_step_ms': phases.mean_ms('m_step'),
        'm_step': m_step_mode(optim), 'clock': clock,
        'roofline': roof,
        'kernels': kern,
    }
    # what an iteration costs besides its big calls (small kernels, the M-step, host gaps)
    out['kernels_ms_per_step'] = sum(k['ms'] * k['launches'] for k in kern.values()) / steps
    out['fixed_ms_per_step'] = out['ms_per_step'] - out['kernels_ms_per_step']
    if shard_of:
        out['shard'] = {'of': shard_of, 'frames': n_local, 'utterances': len(lengths),
                        'note': f'rank 0\'s shard of a {shard_of}-way split of the {datasize}-frame '
                                'corpus on ONE GPU, no collective: value = frames of the shard / '
                                'time; x N is the projection for N GPUs'}
        out['value'] = n_local * steps / elapsed
        # the same iterations recorded as HIP graphs (beer.CapturedIteration: E-step, statistics,
        # update and the phone loop's weight rewrite as ONE submission per iteration)
        ploop_c = make_phone_loop(cov, device)
        it = beer.CapturedIteration(ploop_c, beer.VBConjugateOptimizer(
            ploop_c.mean_field_factorization(), 1.), (X, lengths), datasize=datasize,
            frame_images=images)
        for _ in range(4):
            it()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            it()
        torch.cuda.synchronize()
        out['shard']['captured_ms_per_step'] = 1e3 * (time.perf_counter() - t0) / steps
        out['shard']['captured_mode'] = it.mode
    if with_cpu_baseline and not args.no_cpu_baseline and world == 1 and cov == 'diagonal':
        out['cpu_baseline'] = cpu_baseline_hmm()
    fi = frame_image_report(images)
    if fi:
        out['frame_image'] = fi
    return out


# --------------------------------------------------------------------------------------------
# config 4: HMM-VAE (the prior's "statistics-in" hot path)
# --------------------------------------------------------------------------------------------



def run_vae(args, device, frames=5_000_000, n_minibatches=5, warmup=1):
    '''BASELINE config 4 (`configs[3]`): HMM-VAE, D = 40 frames, residual feed-forward
    encoder / decoder (2 blocks x 128, beer/nnet), 64-dimensional Normal latent, phone-loop
    HMM prior (40 phones x 3 states, one Gaussian per state), ONE EPOCH over the 5 M-frame corpus
    in five minibatches of ~1 M frames (utterances of 200-400 frames).  Per covariance type
    of the prior: the whole VAE step (ELBO + backward to the networks + statistics +
    natural-gradient / Adam update) and the prior's hot path alone -- statistics of the latent
    samples -> per-state log-likelihoods -> forward-backward -> gradient w.r.t. the
    statistics and the samples -> accumulation (beer/models/vae.py:63-89, hmm.py:73-100) --
    with the HIP-event time of every C-ABI call and the roofline of the dominant one.
    Rank 0 of a one-process run only: a minibatch does not shard.'''
    lengths_all = hmm_corpus(frames)
    per = -(-len(lengths_all) // n_minibatches)
    batches = [lengths_all[i:i + per] for i in range(0, len(lengths_all), per)]
    total = sum(lengths_all)
    g = torch.Generator(device=device).manual_seed(4)
    X = torch.randn(total, D, generator=g, device=device)
    Xs = torch.split(X, [sum(b) for b in batches])
    names = ('beer_mixtureset_estep', 'beer_hmm_posteriors_fused', 'beer_hmm_forward_backward',
             'beer_hmm_gather', 'beer_hmm_scatter', 'beer_frames_llh_backward', 'beer_pack_resps',
             'beer_normal_accumulate_packed', 'beer_normal_accumulate',
             # the dense-statistics route (several samples per frame; `dense_route` below)
             'beer_suffstats_mean', 'beer_dense_llh', 'beer_dense_llh_backward',
             'beer_suffstats_backward', 'beer_dense_accumulate', 'beer_softmax_groups',
             'beer_rowdot')
    products = ('beer_mixtureset_estep', 'beer_frames_llh_backward', 'beer_normal_accumulate_packed',
                'beer_normal_accumulate', 'beer_dense_llh', 'beer_dense_llh_backward',
                'beer_dense_accumulate')
    out = {'workload': f'configs[3]: HMM-VAE, D={D}, latent {LATENT}, residual encoder/decoder '
                       f'2x128, phone-loop prior {N_PHONES}x3 states (1 Gaussian per state), one EPOCH '
                       f'over the {total}-frame corpus in {len(batches)} minibatches of '
                       f'~{total // len(batches)} fp32 frames ({len(lengths_all)} utterances), 1 sample '
                       'per frame (the prior runs its frame kernels on the samples; `dense_route`: '
                       'the [T, Q] statistics route that several samples per frame take, ONE '
                       'minibatch)',
           'unit': 'frames/s', 'minibatches': len(batches), 'warmup_minibatches': warmup}
    S = 3 * N_PHONES
    for cov in ('diagonal', 'full'):
        Qz = {'diagonal': 2 * LATENT + 2, 'full': LATENT * LATENT + LATENT + 2}[cov]
        torch.manual_seed(4)
        prior = make_phone_loop(cov, device, dim=LATENT, n_comp=1)
        vae = beer.VAE(prior, beer.nnet.ResidualFeedForwardNet(D, 2, 128),
                       beer.nnet.ResidualFeedForwardNet(LATENT, 2, 128)).to(device)
        cjg = beer.VBConjugateOptimizer(vae.mean_field_factorization(), lrate=.1)
        optim = beer.VBOptimizer(cjg, torch.optim.Adam(vae.parameters(), lr=1e-3))

        def vae_step(b):
            optim.init_step()
            elbo = beer.accumulate_elbo(vae, (Xs[b], batches[b]), datasize=total)
            elbo.backward()
            optim.step()
            return elbo

        Z = torch.randn(total, LATENT, generator=g, device=device)
        Zs = torch.split(Z, [sum(b) for b in batches])

        def prior_path(b, dense=False):
            z = Zs[b].clone().requires_grad_(True)
            stats = beer.kernels.differentiable_stats(z, cov, 1) if dense else \
                beer.kernels.sample_stats(z, cov)
            exp_llh = prior.expected_log_likelihood(stats, utt_lengths=batches[b])
            exp_llh.sum().backward()
            acc = prior.accumulate(stats.detach())
            prior.clear_cache()
            return acc

        sub = {}
        for key, fn, nb in (('vae_step', vae_step, len(batches)),
                            ('prior_hot_path', prior_path, len(batches)),
                            ('dense_route', lambda b: prior_path(b, dense=True), 1)):
            for b in range(warmup):
                fn(b)
            torch.cuda.synchronize()
            with KernelTimer(names) as kt:
                t0 = time.perf_counter()
                for b in range(nb):
                    res = fn(b)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
            nfr = sum(sum(batches[b]) for b in range(nb))
            kern = {}
            for nm in names:
                ms, n = kt.mean_ms(nm)
                if n:
                    kern[nm] = {'ms': ms, 'launches_per_minibatch': n / nb}
                    if nm in products:
                        # one [T, Q] x [Q, S] (or its transpose) product, on either route: 2 T Q S
                        # flop (the gradient w.r.t. the samples: 2 T S D (D + 1), the same count)
                        kern[nm]['tflops'] = 2. * (nfr / nb) * Qz * S / (ms * 1e-3) / 1e12
            sub[key] = {'value': nfr / dt, 'frames': nfr, 'minibatches': nb,
                        'ms_per_minibatch': 1e3 * dt / nb, 'kernels': kern}
            if key == 'vae_step':
                sub[key]['ms_per_epoch'] = 1e3 * dt
                sub[key]['elbo_per_frame_last_minibatch'] = float(res) / (total * len(batches[nb - 1]))
        kern = sub['prior_hot_path']['kernels']
        dom = max(kern, key=lambda nm: kern[nm]['ms'] * kern[nm]['launches_per_minibatch'])
        mb_frames = total / len(batches)
        pmc_key = {'beer_frames_llh_backward': 'c4_sgrad_kernel', 'beer_mixtureset_estep': 'c4_llhx_kernel',
                   'beer_normal_accumulate_packed': 'c4_accx_kernel',
                   'beer_normal_accumulate': 'c4_accd_kernel',
                   'beer_hmm_posteriors_fused': 'c4_fb_wave_kernel',
                   'beer_hmm_forward_backward': 'c4_fb_wave_kernel'}.get(dom)
        if 'tflops' in kern[dom]:
            # float32 [T, Q] products on the bf16 matrix pipes, three pieces per operand: priced
            # like config 2 against the dense bf16 peak
            roof = {'bound': 'mfma', 'kernel': dom, 'achieved': kern[dom]['tflops'],
                    'peak': PEAK_TFLOPS['bf16'], 'unit': 'TFLOP/s',
                    'frac': kern[dom]['tflops'] / PEAK_TFLOPS['bf16'],
                    'avg_launch_ms': kern[dom]['ms'],
                    'note': f'achieved = 2*T*Q*S algorithmic flop of one [T, Q={Qz}] x [Q, S={S}] '
                            'product (no symmetry discount) / HIP-event time; six bf16 MFMAs per '
                            'float32 product'}
        else:
            # the dominant call streams [T, S] arrays: HBM bound; algorithmic bytes = the per-state
            # log-likelihoods in and the posteriors out, once
            byts = {'beer_hmm_posteriors_fused': 4. * mb_frames * S * 2,
                    'beer_hmm_forward_backward': 4. * mb_frames * S * 2}.get(dom, 4. * mb_frames * Qz)
            roof = {'bound': 'hbm', 'kernel': dom, 'achieved': byts / (kern[dom]['ms'] * 1e-3) / 1e9,
                    'peak': PEAK_HBM_GBS, 'unit': 'GB/s',
                    'frac': byts / (kern[dom]['ms'] * 1e-3) / 1e9 / PEAK_HBM_GBS,
                    'avg_launch_ms': kern[dom]['ms'],
                    'note': 'achieved = algorithmic bytes of the dominant call (its [T, S] arrays '
                            'once) / HIP-event time; the kernel also keeps its forward columns '
                            '(fp64) in HBM between the two passes: `traffic`'}
        # (the diagonal prior's path has its own passes, keys c4d_*; older rounds: c4_* only)
        if cov == 'diagonal' and pmc_key and pmc_entry('c4d_' + pmc_key[3:]):
            pmc_key = 'c4d_' + pmc_key[3:]
        roof.update(profiled(pmc_key, roof, per_launch_scale=1.))
        sub['roofline'] = roof
        if not args.no_cpu_baseline:
            sub['cpu_baseline'] = cpu_baseline_vae_prior(cov)
        out[cov] = sub
        del vae, prior, optim, cjg, Z, Zs
        torch.cuda.empty_cache()
    out['value'] = out['diagonal']['vae_step']['value']
    return out


# --------------------------------------------------------------------------------------------
# config 5: the recipe end to end -- features -> training with alignments -> Viterbi align
# --------------------------------------------------------------------------------------------

def run_config5(args, device, hours=None, epochs=5, n_comp=4, cpu_sample=6):
    """BASELINE config 5 (`configs[4]`; flow of recipes/aud/steps/monophone.sh:62-146 and
    recipes/timit_v2/steps/train_hmm.sh:99-143, in memory): synthetic 16 kHz audio ->
    `beer features extract` (MFCC + energy + deltas, 42 dimensions; features/extract.py:60-176)
    -> dataset statistics (cli/dataset.py:38-80) -> monophone model, 40 phones x 3 states x 4
    diagonal Gaussians (recipes/aud/conf/hmm.yml speech units) -> alignment graphs of every
    transcription (hmm/mkaligraph.py:18-39) -> `epochs` x (accumulate with the alignment
    graphs + update; accumulate.py:39-63, update.py:41-62) -> Viterbi alignment of every
    utterance (hmm/decode.py).  Wall-clock per stage with the audio already on the device;
    `value` = frames of the corpus / total wall time."""
    hours = hours or float(os.environ.get('BEER_BENCH_C5_HOURS', 3.))
    rng = np.random.RandomState(5)
    srate, phones = 16000, N_PHONES
    lens, total_s = [], 0.
    while total_s < hours * 3600.:
        n = int(rng.uniform(2., 4.) * srate)
        lens.append(n)
        total_s += n / srate
    # band-limited noise with a slowly varying envelope, int16 like a wav file's samples
    g = torch.Generator(device=device).manual_seed(5)
    walls = {}
    t_all = time.perf_counter()

    def stage(name, t0):
        torch.cuda.synchronize()
        walls[name] = time.perf_counter() - t0

    signals = []
    for n in lens:
        x = torch.randn(n, generator=g, device=device)
        env = 1. + .8 * torch.sin(torch.arange(n, device=device) * (2 * np.pi * 3. / srate))
        signals.append((x * env * 3000.).to(torch.int16))
    torch.cuda.synchronize()
    audio_bytes = 2 * sum(lens)
    # -- features
    t0 = time.perf_counter()
    feats = beer.features.extract(signals, as_numpy=False)          # [T_u, 42] float64, device
    stage('features', t0)
    # -- dataset: float32 frames (cli/dataset.py:33 `.float()`), global mean / variance
    t0 = time.perf_counter()
    lengths = [len(f) for f in feats]
    X = torch.cat(feats).float()
    del feats
    mean, var = X.mean(0), X.var(0)
    stage('dataset', t0)
    total, dim = len(X), X.shape[1]
    # -- model (mkphones / mkphoneloopgraph / mkdecodegraph / mkphoneloop with the recipe's own
    #    configuration: benchlib/recipe.py)
    t0 = time.perf_counter()
    ploop, units = recipe.phone_loop(phones, mean.cpu(), var.cpu(),
                                     conf=recipe.hmm_conf(n_normal_speech=n_comp), seed=5)
    ploop = ploop.to(device)
    stage('model', t0)
    sets = ploop.modelset.original_modelset.modelsets
    n_gauss = sum(len(m) * m.n_comp_per_mixture for m in sets)
    # -- alignment graphs: sil, about one phone per ten frames, sil
    seqs = [['sil'] + [int(v) for v in rng.randint(0, phones, max(2, T // 10 - 2))] + ['sil']
            for T in lengths]
    t0 = time.perf_counter()
    gset = beer.graph.compile_alignments(seqs, units)
    graphs = list(gset)
    gset.device_image(torch.float32)
    stage('alignment_graphs', t0)
    # -- training
    optim = beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.)
    names = ('beer_mixtureset_estep', 'beer_mixtureset_lognorm_image', 'beer_hmm_posteriors_fused',
             'beer_hmm_forward_backward', 'beer_mixtureset_accumulate_fused',
             'beer_normal_accumulate', 'beer_hmm_viterbi', 'beer_hmm_gather')
    elbos = []
    statics = beer.ShardStatics()      # offsets, weights and batch descriptors of the shard: the caller's
    images = beer.FrameImages(X)       # ... and the frames' fragment images (built in the first epoch)
    with KernelTimer(names) as kt:
        t0 = time.perf_counter()
        epoch_s = []
        for _ in range(epochs):
            te = time.perf_counter()
            optim.init_step()
            elbo = beer.accumulate_elbo(ploop, (X, lengths), datasize=total, inference_graphs=graphs,
                                        statics=statics, frame_images=images)
            elbo.backward()
            optim.step()
            elbos.append(elbo.value)
            torch.cuda.synchronize()
            epoch_s.append(time.perf_counter() - te)
        stage('training', t0)
        # -- Viterbi alignment
        t0 = time.perf_counter()
        paths = beer.decode_batch(ploop, (X, lengths), inference_graphs=graphs)
        stage('viterbi_align', t0)
    wall = time.perf_counter() - t_all
    stage_sum = sum(walls.values())
    elbos = [float(v) / (len(lengths) * total) for v in elbos]
    # beside the stages (not in `value`): the same epochs recorded as HIP graphs -- one submission
    # per epoch instead of ~100 launches behind a host that waits for every epoch's ELBO
    captured = None
    if not args.no_extras:
        try:
            it = beer.CapturedIteration(ploop, beer.VBConjugateOptimizer(ploop.mean_field_factorization(), 1.),
                                        (X, lengths), datasize=total, inference_graphs=graphs,
                                        statics=statics, frame_images=images)
            modes = []
            for _ in range(2 * len(ploop.mean_field_factorization()) + 1):
                it()
                modes.append(it.mode)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(epochs):
                it()
            torch.cuda.synchronize()
            captured = {'epoch_ms': 1e3 * (time.perf_counter() - t0) / epochs, 'mode': it.mode,
                        'frames_per_s': total * epochs / (time.perf_counter() - t0)}
        except Exception as err:                           # noqa: BLE001  (an extra, never the line)
            captured = {'error': f'{type(err).__name__}: {err}'[:200]}
    kern = {}
    for nm in names:
        ms, n = kt.mean_ms(nm)
        if n:
            kern[nm] = {'ms': ms, 'launches': n}
    out = {'workload': f'configs[4]: {total_s / 3600.:.2f} h of synthetic 16 kHz audio in {len(lens)} '
                       f'utterances -> {dim}-dimensional MFCC+E+deltas ({total} frames) -> the recipe\'s '
                       f'monophone model (recipes/aud/conf/hmm.yml: 1 non-speech unit x 5 states x 10 + '
                       f'{phones} speech units x 3 states x {n_comp} diagonal Gaussians = {n_gauss}) trained for '
                       f'{epochs} epochs with alignment graphs (~{np.mean([len(q) for q in seqs]):.0f} '
                       'phones per utterance) -> Viterbi alignment; in memory, audio resident on the '
                       'device',
           'unit': 'frames/s', 'value': total / stage_sum, 'wall_s': stage_sum,
           'wall_s_with_synthesis': wall, 'stages_s': walls, 'epoch_s': epoch_s, 'epochs': epochs, 'frames': total,
           'utterances': len(lens), 'audio_bytes': audio_bytes,
           'training_frames_per_s': total * epochs / walls['training'], 'gaussians': n_gauss,
           'captured_epochs': captured,
           'training_frame_gaussians_per_s': total * epochs * n_gauss / walls['training'],
           'features_frames_per_s': total / walls['features'],
           'viterbi_frames_per_s': total / walls['viterbi_align'],
           'elbo_per_frame_by_epoch': elbos, 'elbo_monotone': bool(all(b >= a - 1e-7 * abs(a) for a, b in zip(elbos, elbos[1:]))),
           'aligned_frames': int(sum(len(p) for p in paths)),
           'pcie_note': f'the {audio_bytes / 1e6:.0f} MB of int16 samples would add '
                        f'{audio_bytes / 63e9 * 1e3:.1f} ms over PCIe (63 GB/s) when they start on the host',
           'kernels': kern}
    if not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline_config5(signals, seqs, units, ploop, mean, var, epochs,
                                                   n_comp, total, cpu_sample)
    return out


# --------------------------------------------------------------------------------------------
# config 1: diagonal GMM K = 8, D = 2, 1000 frames -- the latency of ONE iteration
# --------------------------------------------------------------------------------------------

def run_config1(args, device, iters=400):
    """BASELINE config 1 (`configs[0]`: examples/Mixture Model.ipynb): diagonal-covariance
    mixture, K = 8, D = 2, 1000 fp64 frames, the notebook's loop body (init_step,
    evidence_lower_bound, backward, step) as it stands -- an iteration here is launch-bound,
    so the figure is MICROSECONDS PER ITERATION: eager (every kernel launched by the host),
    the library's default (the M-step of a group replayed from its captured graph), and the
    whole iteration as one captured HIP graph (`beer.CapturedIteration`); pipelined (the host
    never waits) and with the ELBO read back after every iteration, as the notebook does."""
    g = torch.Generator().manual_seed(0)
    X = torch.randn(1000, 2, dtype=torch.float64, generator=g)

    def make():
        torch.manual_seed(0)
        ns = beer.NormalSet.create(X.mean(0), X.var(0), size=8, prior_strength=1., noise_std=1.,
                                   cov_type='diagonal')
        return beer.Mixture.create(ns, prior_strength=1.).double().to(device)
    Xd = X.to(device)

    def loop_body(model, optim):
        def body():
            optim.init_step()
            elbo = beer.evidence_lower_bound(model, Xd)
            elbo.backward()
            optim.step()
            return elbo.value
        return body

    def measure(fn):
        for _ in range(10):
            v = fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            v = fn()
        torch.cuda.synchronize()
        piped = (time.perf_counter() - t0) / iters
        t0 = time.perf_counter()
        for _ in range(iters // 4):
            v = float(fn())
        synced = (time.perf_counter() - t0) / (iters // 4)
        return {'us_per_iteration': 1e6 * piped, 'us_per_iteration_elbo_read_back': 1e6 * synced,
                'frames_per_s': 1000 / piped, 'last_elbo': float(v)}
    out = {'workload': 'configs[0]: diagonal GMM K=8, D=2, 1000 fp64 frames, one call of '
                       'evidence_lower_bound + backward + step per iteration (examples/Mixture '
                       'Model.ipynb cell 9)', 'iterations_timed': iters}
    m = make()
    out['eager'] = measure(loop_body(m, beer.VBConjugateOptimizer(m.mean_field_factorization(), 1.,
                                                                  graph=False)))
    m = make()
    out['default'] = measure(loop_body(m, beer.VBConjugateOptimizer(m.mean_field_factorization(), 1.)))
    m = make()
    it = beer.CapturedIteration(m, beer.VBConjugateOptimizer(m.mean_field_factorization(), 1.), Xd)
    out['captured'] = measure(it)
    out['captured']['mode'] = it.mode
    out['value'] = out['captured']['frames_per_s']
    out['unit'] = 'frames/s (1000-frame iterations; see us_per_iteration)'
    if not args.no_cpu_baseline:
        out['cpu_baseline'] = cpu_baseline_config1(X, make)
    return out


def frame_image_report(images):
    '''What the caller-owned frame images (beer_amd.FrameImages: diagonal emissions) hold and
    cost: bytes of images and of the frames the object keeps alive, builds / hits during
    this run, and the time of one build (built in the warm-up iteration, once per block of
    frames, reused for as long as the frames stay where they are -- like the frames
    themselves they are input layout, not model state; BEER_FRAME_IMAGE=0 runs without).'''
    if images is None or not images.builds:
        return None
    X = images.X
    n = min(len(X), 1 << 20)
    probe = beer.FrameStats(X[:n].clone(), 'diagonal')
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    probe.frame_image()
    t1.record()
    torch.cuda.synchronize()
    return {'owner': 'beer_amd.FrameImages(X), passed to accumulate_elbo(frame_images=...)',
            'bytes_held': images.bytes_held, 'frames_bytes': images.frames_bytes,
            'budget_bytes': images.budget, 'builds': images.builds, 'hits': images.hits,
            'build_ms_per_million_frames': t0.elapsed_time(t1) * 1e6 / n,
            'note': 'bf16x3 fragments of phi(x) per 32-frame tile, a function of the frames only: '
                    'built in the warm-up iteration, reused by every timed one'}


def config3_subobject(line):
    'The keys of a config-3 line that go into the default line as a sub-object.'
    keep = ('value', 'unit', 'ms_per_step', 'steps', 'warmup', 'scaling', 'f32_mode', 'kernels',
            'roofline', 'cpu_baseline', 'frame_image', 'count_conservation_rel_err', 'elbo_per_frame',
            'all_reduce_ms', 'm_step_ms', 'm_step', 'frames_per_rank', 'ranks', 'backend',
            'fixed_ms_per_step', 'kernels_ms_per_step', 'shard', 'clock')
    sub = {k: line[k] for k in keep if k in line}
    sub['workload'] = line['config']['workload']
    return sub


def shard_projection(full, shard, n):
    """What the one-GPU measurement of a rank's shard says about an N-GPU run of the same
    corpus: the shard's time against a perfect split of the one-GPU time (the all-reduce
    of the statistics -- 1.3 MB over xGMI -- is not in it)."""
    ideal = full['ms_per_step'] / n
    return {'gpus': n, 'one_gpu_ms_per_step': full['ms_per_step'], 'ideal_ms_per_step': ideal,
            'shard_ms_per_step': shard['ms_per_step'], 'ratio_to_ideal': shard['ms_per_step'] / ideal,
            'speedup': full['ms_per_step'] / shard['ms_per_step'],
            'frames_per_s': full['config']['frames_total'] / (shard['ms_per_step'] * 1e-3)}


# --------------------------------------------------------------------------------------------
# launch
# --------------------------------------------------------------------------------------------

def worker(args):
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', rank))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    # BEER_BENCH_BACKEND=gloo lets the N > 1 code path be exercised on a box with
    # fewer GPUs than ranks (ranks then share devices); the default is RCCL.
    backend = os.environ.get('BEER_BENCH_BACKEND', 'nccl')
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if args.config4_only:
        print(json.dumps(run_vae(args, device)), flush=True)
        return
    if args.config5_only:
        print(json.dumps(run_config5(args, device)), flush=True)
        return
    run = run_gmm if args.config == 2 else run_hmm
    out = run(args, rank, world, device, backend)
    if args.config == 2 and not args.no_config3:
        # BASELINE config 3 inside the same line: diagonal emissions on the 10 M-frame
        # corpus, full covariances on 2 M frames (every rank takes part: the corpus is
        # sharded over them)
        torch.cuda.empty_cache()
        c3 = run_hmm(args, rank, world, device, backend, cov='diagonal', total_frames=10_000_000,
                     steps=6, warmup=2)
        torch.cuda.empty_cache()
        c3f = run_hmm(args, rank, world, device, backend, cov='full', total_frames=2_000_000,
                      steps=6, warmup=2, with_cpu_baseline=False)
        c3s = None
        if world == 1:
            # one rank's share of an 8-GPU run of config 3, on this one GPU
            torch.cuda.empty_cache()
            c3s = run_hmm(args, rank, world, device, backend, cov='diagonal',
                          total_frames=10_000_000, steps=20, warmup=3, with_cpu_baseline=False,
                          shard_of=8)
        if rank == 0:
            out['config3'] = config3_subobject(c3)
            out['config3_full'] = config3_subobject(c3f)
            if c3s:
                out['config3_shard'] = config3_subobject(c3s)
                out['config3_shard']['projected_8gpu'] = shard_projection(c3, c3s, 8)
        if rank == 0 and world == 1 and not args.no_config4:
            torch.cuda.empty_cache()
            out['config4'] = run_vae(args, device)
        if rank == 0 and world == 1 and not args.no_config1:
            out['config1'] = run_config1(args, device)
        if rank == 0 and world == 1 and not args.no_config5:
            torch.cuda.empty_cache()
            out['config5'] = run_config5(args, device)
    if rank == 0:
        if args.config == 2:
            out['summary'] = summary(out)
        else:
            out['summary'] = {'config3_frames_per_s': round(out['value']),
                              'config3_ms_per_step': round(out['ms_per_step'], 3),
                              'config3_roofline_frac': round(out['roofline']['frac'], 4)}
        # stdout carries ONE compact line (< 4 KB: what the driver parses); the full objects go
        # to bench_detail.json and, one JSON line per sub-object, to stderr
        emit(out, ROOT)
    if world > 1:
        dist.destroy_process_group()


def _spawned(rank, args, port):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    worker(args)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', type=int, default=2, choices=(2, 3),
                    help='BASELINE.json config: 2 = GMM K=256 full (headline), 3 = phone-loop HMM')
    ap.add_argument('--frames', type=int, default=None,
                    help='config 2: frames per GPU (1,000,000); config 3: frames of the whole '
                         'corpus (10,000,000)')
    ap.add_argument('--chunk', type=int, default=8192, help='config 2: frames per "utterance"')
    ap.add_argument('--cov', default='diagonal', help='config 3: covariance type of the emissions')
    ap.add_argument('--max-frames', type=int, default=1 << 24,
                    help='config 3: frames per launch of the batched E-step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-exact', action='store_true', help='config 2: skip the f32_exact leg')
    ap.add_argument('--no-check', action='store_true', help='config 2: skip the oracle check')
    ap.add_argument('--no-mstep-graph', action='store_true',
                    help='launch the M-step kernel by kernel instead of replaying its captured HIP graph')
    ap.add_argument('--no-extras', action='store_true',
                    help='profiling runs: no clock probe, no captured variant (their launches carry '
                         'the names of the timed ones)')
    ap.add_argument('--no-config3', action='store_true',
                    help='default line: skip the config3 / config3_full sub-objects')
    ap.add_argument('--no-config4', action='store_true',
                    help='default line: skip the config4 sub-object (HMM-VAE, one process)')
    ap.add_argument('--no-config1', action='store_true',
                    help='default line: skip the config1 sub-object (latency of one small iteration)')
    ap.add_argument('--no-config5', action='store_true',
                    help='default line: skip the config5 sub-object (recipe end to end)')
    ap.add_argument('--config5-only', action='store_true',
                    help='print the config5 sub-object alone')
    ap.add_argument('--config4-only', action='store_true',
                    help='print the config4 sub-object alone (no config 2 / 3 runs)')
    ap.add_argument('--scaling', default='weak', choices=('weak', 'strong'),
                    help='config 2: 1 M frames per GPU (weak, default) or --frames in total (strong)')
    args = ap.parse_args()
    if args.frames is None:
        args.frames = 1_000_000 if args.config == 2 else 10_000_000
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher: spawn the ranks ourselves (one process per GPU, loopback rendezvous)
        import torch.multiprocessing as mp
        with socket.socket() as s:
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
        mp.spawn(_spawned, args=(args, port), nprocs=args.gpus, join=True)
        return
    worker(args)


if __name__ == '__main__':
    main()
