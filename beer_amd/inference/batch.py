"""Batched E-step: the MI355X-first counterpart of the reference's
per-utterance accumulation loop.

The reference's data-parallel "map" step (`beer hmm accumulate`,
beer/cli/subcommands/hmm/accumulate.py:39-59) calls `evidence_lower_bound`
once per utterance -- a Python loop over ~300-frame utterances with tens of
torch ops each.  `accumulate_elbo` does the same work for a whole shard in a
handful of kernel launches over a ragged batch and returns the SAME object the
loop would have produced:

    elbo = evidence_lower_bound(datasize=N)
    for utt in shard:
        elbo += evidence_lower_bound(model, utt, datasize=N, ...)

i.e. value = sum_u (N / T_u) sum_t l_ut - U * KL(q || p)   (quirk Q1),
acc_stats = sum_u acc_u, minibatchsize = sum_u T_u.  The KL term is computed
once.  Sufficient statistics are accumulated in fp64 across the whole shard.
"""

import os
import threading

import torch

from .. import _hip, hmm_kernels as hk, kernels
from ..models.gaussians import NormalSet
from ..models.mixtures import Mixture, MixtureSet
from ..models.modelset import JointModelSet
from ..models.sequence import HMM, PhoneLoop
from ..models.vae import VAE
from ..models.weights import SBCategorical
from ..stats import FrameStats
from .objectives import EvidenceLowerBoundInstance

__all__ = ['accumulate_elbo', 'pack_utterances', 'decode_batch', 'ShardStatics']

# Scratch of one sub-batch (responsibilities, per-state likelihoods, trellis): up to
# three sub-batches are in flight, so a sub-batch may take an eighth of what is free on
# the device when the first batch is cut, at most 48 GB (a 288 GB MI355X to itself:
# 36 GB -- the 10 M frames of BASELINE config 3 are then ONE launch of every kernel, 29 GB
# of scratch; a smaller or shared device: less).  BEER_SCRATCH_GB overrides.
_SCRATCH_CAP = 48 << 30
_scratch_bytes = [int(os.environ['BEER_SCRATCH_GB']) << 30 if 'BEER_SCRATCH_GB' in os.environ
                  else None]


def _scratch_budget():
    if _scratch_bytes[0] is None:
        try:
            free, _ = torch.cuda.mem_get_info()
            _scratch_bytes[0] = max(1 << 28, min(_SCRATCH_CAP, free // 8))
        except Exception:                                   # no device: the caller raises later
            _scratch_bytes[0] = _SCRATCH_CAP
    return _scratch_bytes[0]


class ShardStatics:
    '''What `accumulate_elbo` derives from the utterance LENGTHS and the data-set size alone
    -- offsets, the per-utterance weights datasize / T_u on the device, the cut into
    sub-batches -- kept by the caller across the iterations over one shard (like
    `FrameImages` for the frames: nothing is cached behind the caller's back).  One object
    per (utterances, datasize) pair; a call whose lengths do not match what the object was
    filled with refills it.  Without one these are rebuilt per call: a 33 k-element host
    loop and two small uploads per iteration, and -- uploads cannot be recorded -- no
    capture of the iteration as a HIP graph (`CapturedIteration`).'''

    def __init__(self):
        self.key, self.items, self.lengths, self.mark = None, {}, None, None

    def bind(self, lengths):
        '''The shard these statics belong to: the same list object (the usual case: O(1)) or
        an equal one keeps them, anything else empties them.'''
        n = len(lengths)
        mark = (n, lengths[0], lengths[n // 2], lengths[-1]) if n else (0,)
        if self.lengths is not lengths or self.mark != mark:
            # (the same list edited in place is caught where it is cheap to look: its length
            #  and three of its entries)
            if self.mark != mark or self.lengths != lengths:
                self.key, self.items = None, {}
            self.lengths, self.mark = lengths, mark

    def entry(self, key, name, make):
        if self.key != key:
            self.key, self.items = key, {}
        if name not in self.items:
            self.items[name] = make()
        return self.items[name]


def _capturing():
    return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()


def pack_utterances(utterances):
    '''(X [sum T_u, D] on the GPU, lengths list).  Accepts a list of [T_u, D]
    tensors or an already packed `(X, lengths)` pair.'''
    if isinstance(utterances, tuple) and len(utterances) == 2 and \
            isinstance(utterances[0], torch.Tensor):
        X, lengths = utterances
        # (a list of Python ints is handed through as it is: `ShardStatics` recognises the
        # caller's list by identity)
        if not (isinstance(lengths, list) and set(map(type, lengths)) <= {int}):
            lengths = [int(n) for n in lengths]
        return _hip.on_device(X), lengths
    utterances = list(utterances)
    lengths = [len(u) for u in utterances]
    dev = _hip.require_device()
    X = torch.cat([u.to(dev) for u in utterances], dim=0).contiguous()
    return X, lengths


def _groups(emissions):
    'Flatten an emission model into [(MixtureSet | NormalSet, S, G)].'
    if isinstance(emissions, JointModelSet):
        out = []
        for m in emissions.modelsets:
            out += _groups(m)
        return out
    if isinstance(emissions, MixtureSet):
        if not isinstance(emissions.modelset, NormalSet):
            raise NotImplementedError('MixtureSet components must be a NormalSet')
        return [(emissions, len(emissions), emissions.n_comp_per_mixture)]
    if isinstance(emissions, NormalSet):
        return [(emissions, len(emissions), 1)]
    raise NotImplementedError(f'unsupported emission model {type(emissions).__name__}')


def _normalset(group):
    return group.modelset if isinstance(group, MixtureSet) else group


def _sub_batches(lengths, bytes_per_frame, max_frames):
    '''Split utterance indices into runs whose scratch fits the budget.  The
    runs are balanced (same number of frames, within an utterance) rather than
    filled greedily: equal-sized scratch tensors are recycled by the caching
    allocator, unequal ones make it hipMalloc / hipFree gigabytes every
    iteration (25 ms stalls at config 3).'''
    budget = max(1, min(max_frames, _scratch_budget() // max(1, bytes_per_frame)))
    total = sum(lengths)
    n_runs = max(1, -(-total // budget))
    target = -(-total // n_runs)
    runs, cur, n = [], [], 0
    for u, T in enumerate(lengths):
        if cur and (n + T > budget or (n >= target and len(runs) < n_runs - 1)):
            runs.append(cur)
            cur, n = [], 0
        cur.append(u)
        n += T
    if cur:
        runs.append(cur)
    return runs


# Per host thread (a thread drives its own stream of sub-batches: two training loops in two
# threads must not pop each other's events) -- the throttle's queue and the KL side streams.
_local = threading.local()


def _throttle(depth=2):
    '''Keep the host at most `depth` sub-batches ahead of the GPU.  Every
    sub-batch allocates gigabytes of scratch (responsibilities, per-state
    likelihoods); a host that queues many of them before the first has run
    makes the caching allocator hipMalloc new blocks instead of recycling --
    tens of ms each, and the stream waits for them.'''
    if _capturing():
        return torch.cuda.Event()             # (a recording does not run: nothing to wait for)
    in_flight = _local.__dict__.setdefault('in_flight', [])
    while len(in_flight) >= depth:
        in_flight.pop(0).synchronize()
    ev = torch.cuda.Event()
    in_flight.append(ev)
    return ev


def _cached_batch(graph, run_lengths, dtype):
    '''Batch descriptor of utterances that all use `graph` (the free phone
    loop).  Training iterates over the same shard again and again: the
    descriptor is kept on the graph and reused as long as the graph's device
    image is the same object (its weights are refreshed in place), so the
    steady-state iteration has no host -> device copy at all.'''
    cache = graph.__dict__.setdefault('_batch_cache', {})
    key = (dtype, tuple(run_lengths))
    dg = graph.device_graph(dtype)
    hit = cache.get(key)
    if hit is not None and hit.dgraphs[0] is dg:
        hit.struct.all_lowdeg = hit.lowdeg_default       # forward_backward may have cleared it
        return hit
    if len(cache) >= 8:
        cache.clear()
    batch = hk.HmmBatch([graph], [0] * len(run_lengths), run_lengths, dtype)
    batch.lowdeg_default = batch.struct.all_lowdeg
    cache[key] = batch
    return batch


_KL_BESIDE = os.environ.get('BEER_KL_SIDE_STREAM', '1') != '0'


def _kl_beside_the_estep(model, device):
    '''KL(q || p) of the model's parameters on a side stream: a dozen small launches
    (expected statistics and log-normalisers of the Dirichlets, the KL kernels, their
    sums) that depend on the parameters only, not on the frames -- queued beside the
    E-step kernels instead of in front of them.  Returns (kl, event or None); the
    caller makes its stream wait for the event before it uses `kl`.  The side stream
    starts behind everything queued so far (the previous M-step), and the next call
    starts behind this one's consumers: tensors it allocates are not recycled early.'''
    if not _KL_BESIDE or isinstance(model, VAE) or device.type != 'cuda' or _capturing():
        return torch.as_tensor(model.kl_div_posterior_prior()), None
    main = torch.cuda.current_stream(device)
    # Everything the distributions memoise -- expected statistics (which the E-step
    # reads: `natural_form()`), log-normalisers, natural parameters -- is produced HERE,
    # on the main stream: a memo entry written by a side-stream kernel would be picked up
    # by the main stream's E-step with nothing ordering the two.  What is left for the
    # side stream are the KL kernels and their sums, whose results nobody reads before
    # the caller has waited for the event.
    for param in model.bayesian_parameters():
        for dist in (getattr(param, 'posterior', None), getattr(param, 'prior', None)):
            if dist is not None and hasattr(dist, 'expected_sufficient_statistics'):
                dist.expected_sufficient_statistics()
                dist.log_norm()
                dist.natural_parameters()
    side_streams = _local.__dict__.setdefault('side_streams', {})
    side = side_streams.get(device)
    if side is None:
        side = side_streams[device] = torch.cuda.Stream(device)
    side.wait_stream(main)
    with torch.cuda.stream(side):
        kl = torch.as_tensor(model.kl_div_posterior_prior())
        done = side.record_event()
    return kl, done


def _finish(model, value_terms, kl, nutt, acc, datasize, total_frames):
    value = value_terms - float(nutt) * kl.to(value_terms.device, torch.float64)
    return EvidenceLowerBoundInstance(value, acc, model.bayesian_parameters(),
                                      total_frames, datasize)


def _like(param, t):
    return t.to(dtype=param.stats.dtype, device=param.stats.device)


def _mixture_batch(model, X, lengths, datasize, labels, max_frames, statics=None):
    ns = model.modelset
    K, cov = len(ns), ns.cov_type
    dev, dtype = X.device, X.dtype
    exp_T = ns.means_precisions.natural_form()
    lw = model._log_weights().view(1, K)
    Q = FrameStats(X[:1], cov).shape[1]
    acc = torch.zeros(K, Q, dtype=torch.float64, device=dev)
    utt_llh = torch.zeros(len(lengths), dtype=torch.float64, device=dev)

    def offsets_and_scales():
        off = torch.zeros(len(lengths) + 1, dtype=torch.int64)
        off[1:] = torch.cumsum(torch.as_tensor(lengths, dtype=torch.int64), 0)
        # every host -> device copy happens here, BEFORE the first big kernel: a copy
        # from pageable memory makes the host wait for the stream, and one issued
        # after the E-step kernels would keep the host from queueing the M-step
        # launches while those kernels run
        up = _hip.upload({'off': off, 'scales': torch.as_tensor(
            [datasize / float(T) for T in lengths], dtype=torch.float64)}, dev)
        return off, up['off'], up['scales']
    statics = statics if statics is not None else ShardStatics()
    statics.bind(lengths)
    skey = ('mixture', len(lengths), float(datasize), str(dev))
    off, off_dev, scales = statics.entry(skey, 'offsets', offsets_and_scales)
    lab_dev = None if labels is None else \
        _hip.on_device(torch.as_tensor(labels)).to(torch.int64).contiguous()
    for run in _sub_batches(lengths, K * X.element_size(), max_frames):
        done = _throttle()
        f0, f1 = int(off[run[0]]), int(off[run[-1] + 1])
        stats = FrameStats(X[f0:f1], cov)
        lab = None if lab_dev is None else lab_dev[f0:f1]
        wide = kernels.wide_mixture_split(stats, K, cov) if lab is None else None
        if lab is None and kernels.packed_path_ok(stats, K, cov):
            # responsibilities go to the accumulation already split for its fp16 products
            log_norm, resps = kernels.mixture_estep_packed(stats, exp_T, lw, K, cov)
        elif wide:
            # K > 256: blocks of components on the mixture-set kernels, two-level softmax
            log_norm, resps = kernels.wide_mixture_estep(stats, exp_T, lw, K, cov, wide)
        else:
            log_norm, resps = kernels.mixtureset_estep(stats, exp_T, lw, 1, K, cov, labels=lab)
        seg = off_dev[run[0]:run[-1] + 2] - f0
        hk.segment_sum(log_norm.view(-1), seg, len(run), out=utt_llh[run[0]:run[-1] + 1])
        kernels.normal_accumulate(stats, resps, None, K, 1, cov, acc=acc)
        done.record()
    value_terms = (scales * utt_llh).sum()
    wparam = model.categorical.mean_field_factorization()[0][0]
    if isinstance(model.categorical, SBCategorical):
        wacc = -2. * acc[:, -2]
    else:
        wacc = kernels.weights_from_acc(acc, 1, K).view(-1)
    out = {wparam: _like(wparam, wacc), ns.means_precisions: _like(ns.means_precisions, acc)}
    return value_terms, out


def _emission_estep(groups, stats, dtype, for_accumulate=False):
    '''pc_all [T, S_total] + per group what its accumulation needs: the component
    responsibilities [T, S*G] (float32 matrix, or `PackedResps` where
    `kernels.packed_sets_ok`), or -- where the accumulation recomputes them from
    the frames (`kernels.fused_accumulate_ok`) -- the group's log-normalisers.'''
    cols, comps = [], []
    for grp, S, G in groups:
        ns = _normalset(grp)
        lw = grp._log_weights() if isinstance(grp, MixtureSet) else None
        gstats = stats.as_cov(ns.cov_type)
        fused = for_accumulate and G > 1 and \
            kernels.fused_accumulate_ok(gstats, S, G, ns.cov_type)
        if for_accumulate and G > 1 and not fused and \
                kernels.packed_sets_ok(gstats, S, G, ns.cov_type):
            # full covariance: responsibilities as the accumulation kernel's tiles
            log_norm, packed = kernels.mixtureset_estep_packed(
                gstats, ns.means_precisions.natural_form(), lw, S, G, ns.cov_type)
            cols.append(log_norm)
            comps.append(packed)
            continue
        log_norm, resps = kernels.mixtureset_estep(
            gstats, ns.means_precisions.natural_form(), lw, S, G, ns.cov_type,
            want_resps=for_accumulate and G > 1 and not fused)
        cols.append(log_norm)
        comps.append(('fused', log_norm, lw) if fused else resps)
    pc_all = cols[0] if len(cols) == 1 else torch.cat(cols, dim=-1)
    return pc_all, comps


def _hmm_batch(model, X, lengths, datasize, graphs, scale, viterbi, state_paths, max_frames,
               frame_images=None, statics=None):
    emissions = model._emissions()
    groups = _groups(emissions)
    S_total = sum(S for _, S, _ in groups)
    K_max = sum(S * G for _, S, G in groups)
    dev, dtype = X.device, X.dtype
    free_loop = graphs is None
    accs = []
    for grp, S, G in groups:
        ns = _normalset(grp)
        Q = FrameStats(X[:1], ns.cov_type).shape[1]
        accs.append(torch.zeros(S * G, Q, dtype=torch.float64, device=dev))
    nutt = len(lengths)
    utt_llh = torch.zeros(nutt, dtype=torch.float64, device=dev)

    def offsets_and_scales():
        off = torch.zeros(nutt + 1, dtype=torch.int64)
        off[1:] = torch.cumsum(torch.as_tensor(lengths, dtype=torch.int64), 0)
        return off, _hip.to_device(torch.as_tensor([datasize / float(T) for T in lengths],
                                                   dtype=torch.float64), dev)
    statics = statics if statics is not None else ShardStatics()
    statics.bind(lengths)
    skey = ('hmm', nutt, float(datasize), str(dev))
    off, scales = statics.entry(skey, 'offsets', offsets_and_scales)
    xi_tot = g0_tot = flow_tot = None
    max_S = model.graph.n_states if free_loop else max(g.n_states for g in graphs)
    # scratch per frame: the responsibilities of the groups whose accumulation does
    # not recompute them, per-state likelihoods / posteriors, the trellis
    K_scratch = sum(S * G for grp, S, G in groups if G > 1 and not kernels.fused_accumulate_ok(
        FrameStats(X, _normalset(grp).cov_type), S, G, _normalset(grp).cov_type))
    bpf = (K_scratch + 2 * S_total) * X.element_size() + max_S * (3 * X.element_size() + 8)
    if frame_images is None or not frame_images.covers(X):
        # frame fragment images built per sub-batch and freed with it (the caller keeps none)
        bpf += max((_hip.lib().beer_frame_image_bytes(
            _hip.COV_CODE[_normalset(grp).cov_type], 1 << 20, X.shape[1]) >> 20)
            for grp, _, _ in groups) if X.dtype == torch.float32 else 0
    runs = statics.entry(skey, ('runs', bpf, max_frames), lambda: [
        (run, int(off[run[0]]), int(off[run[-1] + 1]), [lengths[u] for u in run])
        for run in _sub_batches(lengths, bpf, max_frames)])
    for run_index, (run, f0, f1, run_lengths) in enumerate(runs):
        done = _throttle()
        # the emission E-step is queued first: building the batch descriptor (host
        # work + one asynchronous copy from pinned memory) overlaps with it
        stats = FrameStats(X[f0:f1], _normalset(groups[0][0]).cov_type, images=frame_images)
        pc_all, comps = _emission_estep(groups, stats, dtype, for_accumulate=True)
        if free_loop:
            batch = _cached_batch(model.graph, run_lengths, dtype)
        else:
            def images_of(uniq):
                """What a cached descriptor's struct pointers point into: one blob per GraphSet
                (alignment graphs compiled together share it), the memoised device image of every
                other graph (rebuilt -- a NEW object -- when its tensors are replaced or rewritten)."""
                sets, out = {}, []
                for g in uniq:
                    owner = getattr(g, '_set', None)
                    if owner is None:
                        out.append(g.device_graph(dtype))
                    elif id(owner) not in sets:
                        sets[id(owner)] = True
                        out.append(owner.device_image(dtype)[0])
                return out

            def make_batch():
                uniq, ids, seen = [], [], {}
                for u in run:
                    g = graphs[u]
                    if id(g) not in seen:
                        seen[id(g)] = len(uniq)
                        uniq.append(g)
                    ids.append(seen[id(g)])
                return hk.HmmBatch(uniq, ids, run_lengths, dtype), graphs, uniq, images_of(uniq)
            # (the descriptor of a run's alignment graphs -- thousands of graph structs, one
            # upload -- is the caller's to keep with the shard.  Its key names everything the cut
            # into runs depends on (bytes per frame, max_frames) and the list of graphs (identity,
            # length, the run's first and last graph); a hit is still checked against the device
            # images its struct pointers were taken from)
            bkey = ('batch', run_index, bpf, max_frames, id(graphs), len(graphs),
                    id(graphs[run[0]]), id(graphs[run[-1]]), str(dtype))
            batch, _, uniq, images = statics.entry(skey, bkey, make_batch)
            if not all(a is b for a, b in zip(images, images_of(uniq))):
                statics.items.pop(bkey)
                batch, _, uniq, images = statics.entry(skey, bkey, make_batch)
        hard = viterbi or state_paths is not None
        # phone counts come from the flows through the loop's hub (the eliminated
        # pivot); a loop whose end -> start arcs stayed ordinary arcs needs xi
        need_counts = free_loop and isinstance(model, PhoneLoop)
        hubbed = need_counts and getattr(getattr(batch.dgraphs[0], 'lowdeg', None), 'n_hubs', 0) >= 1
        if not hard and hk.fused_ok(batch) and (hubbed or not need_counts):
            # gather + forward-backward + scatter in one launch, one wave per utterance
            sr, g0, flow = hk.posteriors_fused(batch, pc_all, scale, want_counts=need_counts,
                                               utt_llh=utt_llh[run[0]:run[-1] + 1])
            xi = None
        else:
            pc_llhs = hk.gather(batch, pc_all, scale)
            if hard:
                if state_paths is None:
                    path = hk.viterbi(batch, pc_llhs)
                else:
                    path = torch.cat([torch.as_tensor(state_paths[u]).reshape(-1) for u in run])
                gamma, xi, g0 = hk.path_posteriors(batch, path, want_xi=free_loop)
                flow = None
            else:
                gamma, xi, g0, _, flow = hk.forward_backward(batch, pc_llhs, want_xi=free_loop)
            sr, _ = hk.scatter(batch, pc_llhs, gamma, S_total, scale, want_exp_llh=False,
                               utt_llh=utt_llh[run[0]:run[-1] + 1])
        if free_loop:
            if xi is not None:
                xi_tot = xi if xi_tot is None else xi_tot + xi
            if g0 is not None:
                g0_tot = g0 if g0_tot is None else g0_tot + g0
            if flow is not None:
                flow_tot = flow if flow_tot is None else flow_tot + flow
        first = 0
        for (grp, S, G), comp, acc in zip(groups, comps, accs):
            ns = _normalset(grp)
            sr_g = sr if len(groups) == 1 else sr[:, first:first + S].contiguous()
            first += S
            gstats = stats.as_cov(ns.cov_type)
            if G == 1:
                kernels.normal_accumulate(gstats, sr_g, None, S, 1, ns.cov_type, acc=acc)
            elif isinstance(comp, tuple):
                # responsibilities recomputed inside the accumulation: no [T, K] matrix
                kernels.mixtureset_accumulate_fused(
                    gstats, ns.means_precisions.natural_form(), comp[2], comp[1], sr_g, S, G,
                    ns.cov_type, acc=acc)
            else:
                kernels.normal_accumulate(gstats, comp, sr_g, S, G, ns.cov_type, acc=acc)
        done.record()
    value_terms = (scales * utt_llh).sum()
    out = {}
    for (grp, S, G), acc in zip(groups, accs):
        ns = _normalset(grp)
        out[ns.means_precisions] = _like(ns.means_precisions, acc)
        if isinstance(grp, MixtureSet):
            wparam = grp.categoricalset.weights
            out[wparam] = _like(wparam, kernels.weights_from_acc(acc, S, G))
    if isinstance(model, PhoneLoop):
        wparam = model.categorical.mean_field_factorization()[0][0]
        ref = wparam.stats
        if free_loop:
            counts = model.phone_counts(xi_tot, g0_tot, flow_tot).to(dtype=ref.dtype,
                                                                     device=ref.device)
            # per-utterance `sufficient_statistics` (last <- sum) then sum over
            # utterances == the same map applied to the summed counts.
            cstats = model.categorical.sufficient_statistics(counts.view(1, -1))
            out.update(model.categorical.accumulate(cstats))
        else:
            fake = torch.zeros(len(model.start_pdf), dtype=ref.dtype, device=ref.device)
            out.update(model.categorical.accumulate(fake[None, :]))
    return value_terms, out


def _vae_batch(model, X, lengths, datasize, nsamples, llh_weight, kl_weight):
    '''One minibatch of utterances through a VAE: the encoder / decoder see
    the packed frames, an HMM prior sees them as a ragged batch.  The value
    keeps its autograd graph (`elbo.backward()` reaches the networks).'''
    kwargs = {'utt_lengths': lengths} if isinstance(model.prior, HMM) else {}
    broadcast, model.reference_broadcast = model.reference_broadcast, False
    try:
        per_frame = model.expected_log_likelihood(X, nsamples=nsamples, llh_weight=llh_weight,
                                                  kl_weight=kl_weight, **kwargs)
    finally:
        model.reference_broadcast = broadcast
    # frame weights datasize / T_u (times T_u with the reference's [T, T] quirk)
    w = [float(datasize) if broadcast else datasize / float(T) for T in lengths]
    weights = torch.repeat_interleave(
        torch.as_tensor(w, dtype=torch.float64, device=per_frame.device),
        torch.as_tensor(lengths, device=per_frame.device))
    value_terms = (weights * per_frame.to(torch.float64)).sum()
    return value_terms, model.accumulate(None)


def accumulate_elbo(model, utterances, datasize=-1, inference_graphs=None, scale=1.,
                    viterbi=False, state_paths=None, labels=None, max_frames=1 << 24,
                    nsamples=1, llh_weight=1., kl_weight=1., frame_images=None, statics=None):
    '''ELBO + accumulated statistics of a shard of utterances, identical to the
    sum of per-utterance `evidence_lower_bound(model, utt, datasize=datasize,
    inference_graph=..., scale=..., viterbi=...)` calls.

    Args:
        model: `Mixture`, `HMM` or `PhoneLoop`.
        utterances: list of [T_u, D] tensors, or `(X_packed, lengths)`.
        datasize: frames in the whole training set (<= 0: this shard).
        inference_graphs: optional list of per-utterance `CompiledGraph`
            (alignment graphs); None = the model's own graph.
        scale: acoustic scale (HMM only).
        viterbi / state_paths: hard-alignment training branches (HMM only).
        labels: optional int64 [sum T_u] component labels (Mixture only).
        nsamples, llh_weight, kl_weight: `VAE.expected_log_likelihood`
            arguments (VAE only; the batch is one minibatch, its value keeps
            the autograd graph of the networks).
        frame_images: optional `FrameImages` of the packed frames (HMM only; the
            caller's, kept across iterations over the same shard).  Without it the
            images of a sub-batch live for this call only.
        statics: optional `ShardStatics` (Mixture, HMM): what depends on the utterance
            lengths and `datasize` only, kept by the caller across iterations.
    '''
    X, lengths = pack_utterances(utterances)
    if any(T <= 0 for T in lengths):
        raise ValueError('empty utterance in the batch')
    total = sum(lengths)
    if datasize <= 0:
        datasize = total
    if len(lengths) == 0:
        return EvidenceLowerBoundInstance(0., {}, [], 0, datasize)
    kl, kl_done = _kl_beside_the_estep(model, X.device)
    if isinstance(model, Mixture):
        value_terms, acc = _mixture_batch(model, X, lengths, datasize, labels, max_frames, statics)
    elif isinstance(model, HMM):
        value_terms, acc = _hmm_batch(model, X, lengths, datasize, inference_graphs, scale,
                                      viterbi, state_paths, max_frames, frame_images, statics)
    elif isinstance(model, VAE):
        value_terms, acc = _vae_batch(model, X, lengths, datasize, nsamples, llh_weight,
                                      kl_weight)
    else:
        raise NotImplementedError(f'no batched E-step for {type(model).__name__}')
    model.clear_cache()
    if kl_done is not None:
        torch.cuda.current_stream(X.device).wait_event(kl_done)
    return _finish(model, value_terms, kl, len(lengths), acc, datasize, total)


def decode_batch(model, utterances, inference_graphs=None, scale=1., max_frames=1 << 20):
    '''Viterbi pdf-id paths for a shard: list of int64 tensors, one per
    utterance (`HMM.decode`, beer/models/hmm.py:105-114, batched).'''
    X, lengths = pack_utterances(utterances)
    groups = _groups(model._emissions())
    S_total = sum(S for _, S, _ in groups)
    K_max = sum(S * G for _, S, G in groups)
    off = [0]
    for T in lengths:
        off.append(off[-1] + T)
    paths = []
    max_S = model.graph.n_states if inference_graphs is None \
        else max(g.n_states for g in inference_graphs)
    bpf = (K_max + S_total) * X.element_size() + max_S * (X.element_size() + 4)
    for run in _sub_batches(lengths, bpf, max_frames):
        f0, f1 = off[run[0]], off[run[-1] + 1]
        stats = FrameStats(X[f0:f1], _normalset(groups[0][0]).cov_type)
        pc_all, _ = _emission_estep(groups, stats, X.dtype)
        run_lengths = [lengths[u] for u in run]
        if inference_graphs is None:
            batch = hk.HmmBatch([model.graph], [0] * len(run), run_lengths, X.dtype)
        else:
            uniq, ids, seen = [], [], {}
            for u in run:
                g = inference_graphs[u]
                if id(g) not in seen:
                    seen[id(g)] = len(uniq)
                    uniq.append(g)
                ids.append(seen[id(g)])
            batch = hk.HmmBatch(uniq, ids, run_lengths, X.dtype)
        path = hk.viterbi(batch, hk.gather(batch, pc_all, scale), map_pdf=True)
        paths += list(torch.split(path, run_lengths))
    return paths
