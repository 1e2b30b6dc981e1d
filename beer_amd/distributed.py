"""Data-parallel VB training: one process per GPU, utterances sharded across
ranks, ONE all-reduce per VB iteration.

The reference's data parallelism is a filesystem map-reduce: N `beer hmm
accumulate` processes pickle `(elbo, count)` tuples and one `beer hmm update`
process sums them with `+` (beer/cli/subcommands/hmm/accumulate.py:62-63,
update.py:41-62).  Here the `+` is a single `all_reduce(SUM)` (RCCL over xGMI
on a GPU node, gloo on CPU for the tests) over one flat fp64 buffer holding
every parameter's accumulated statistics, the ELBO value and the frame /
utterance counts; every rank then applies the identical M-step to identical
inputs (no broadcast needed).  The message is small (3.4 MB fp64 at K = 256,
D = 40 full covariance), so the exchange is latency-bound: what matters is
doing exactly one collective, not one per parameter.
"""

import torch
import torch.distributed as dist

from .inference.objectives import EvidenceLowerBoundInstance

__all__ = ['all_reduce_elbo', 'shard_utterances', 'flatten_elbo', 'unflatten_elbo']


def shard_utterances(lengths, world_size, rank):
    '''Indices of the utterances of `rank`, balanced by FRAME count (longest
    processing time first), not by line count as the reference's `split -n
    l/N` does (recipes/aud/utils/parallel/split.sh:20).'''
    order = sorted(range(len(lengths)), key=lambda u: (-lengths[u], u))
    loads = [0] * world_size
    mine = []
    for u in order:
        r = min(range(world_size), key=lambda i: (loads[i], i))
        loads[r] += lengths[u]
        if r == rank:
            mine.append(u)
    return sorted(mine)


def flatten_elbo(elbo, params, n_utts, device):
    '''[value, minibatchsize, n_utts, stats of params[0], stats of params[1], ...]
    as one fp64 vector.  Missing statistics count as zeros.'''
    # (counts go in by fill kernels: a tensor built from Python numbers would be a
    # copy from pageable memory, which makes the host wait for the stream)
    def count(v):
        if isinstance(v, torch.Tensor):
            return v.to(device=device, dtype=torch.float64).reshape(1)
        return torch.full((1,), float(v), dtype=torch.float64, device=device)
    parts = [torch.as_tensor(elbo.value, dtype=torch.float64, device=device).reshape(1),
             count(elbo._minibatchsize), count(n_utts)]
    for p in params:
        s = elbo._acc_stats.get(p)
        if s is None:
            s = torch.zeros_like(p.stats)
        parts.append(s.to(device=device, dtype=torch.float64).reshape(-1))
    return torch.cat(parts)


def unflatten_elbo(flat, params, datasize):
    value = flat[0].clone()
    if flat.is_cuda:
        # stay on the device: reading the counts back would stall the host until
        # the collective is over, and with it the queueing of the M-step
        mbsize, n_utts = flat[1], flat[2]
    else:
        mbsize, n_utts = int(round(float(flat[1]))), int(round(float(flat[2])))
    acc, first = {}, 3
    for p in params:
        n = p.stats.numel()
        acc[p] = flat[first:first + n].reshape(p.stats.shape).to(dtype=p.stats.dtype,
                                                                   device=p.stats.device)
        first += n
    return EvidenceLowerBoundInstance(value, acc, params, mbsize, datasize), n_utts


def all_reduce_elbo(elbo, model, n_utts=0, group=None):
    '''Sum the ELBO objects of every rank.  Returns (global elbo, global
    utterance count).  With no initialised process group this is the
    identity (single GPU).  Over RCCL the counts (minibatch size of the returned
    ELBO, utterance count) are 0-dim device tensors -- no host synchronisation
    between the E-step and the M-step; `int(...)` them where a number is needed.'''
    params = list(model.bayesian_parameters())
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return elbo, n_utts
    backend = dist.get_backend(group)
    device = torch.device('cuda', torch.cuda.current_device()) if backend == 'nccl' \
        else torch.device('cpu')
    flat = flatten_elbo(elbo, params, n_utts, device)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return unflatten_elbo(flat, params, elbo._datasize)
