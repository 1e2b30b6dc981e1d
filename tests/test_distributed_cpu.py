"""world_size-2 gloo test of the data-parallel reduce (CPU only): the
all-reduced ELBO object equals the `+` of the per-rank objects, and shards
are balanced by frame count."""

import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import beer_amd as beer
from beer_amd.distributed import all_reduce_elbo, flatten_elbo, shard_utterances, unflatten_elbo


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(0)
    X = torch.randn(40, 3, dtype=torch.float64)
    ns = beer.NormalSet.create(X.mean(0), X.var(0), size=4, cov_type='full')
    return beer.Mixture.create(ns)


def _fake_elbo(model, rank):
    'Deterministic per-rank ELBO object (no GPU needed).'
    acc = {}
    for i, p in enumerate(model.bayesian_parameters()):
        acc[p] = torch.full_like(p.stats, float(rank + 1) * (i + 1))
    return beer.EvidenceLowerBoundInstance(torch.tensor(-10. * (rank + 1), dtype=torch.float64),
                                           acc, model.bayesian_parameters(), 100 * (rank + 1),
                                           1000)


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    model = _model()
    elbo, n = all_reduce_elbo(_fake_elbo(model, rank), model, n_utts=rank + 3)
    params = list(model.bayesian_parameters())
    out.put((rank, float(elbo.value), elbo._minibatchsize, n,
             [elbo._acc_stats[p].sum().item() for p in params]))
    dist.destroy_process_group()


def test_all_reduce_matches_sum_of_rank_objects():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    model = _model()
    expect = _fake_elbo(model, 0) + _fake_elbo(model, 1)
    params = list(model.bayesian_parameters())
    sums = [expect._acc_stats[p].sum().item() for p in params]
    for rank, value, mb, n, got in results:
        assert value == float(expect.value) == -30.
        assert mb == expect._minibatchsize == 300
        assert n == 3 + 4
        assert got == sums


def test_flatten_roundtrip_single_process():
    model = _model()
    params = list(model.bayesian_parameters())
    e = _fake_elbo(model, 1)
    flat = flatten_elbo(e, params, 5, torch.device('cpu'))
    back, n = unflatten_elbo(flat, params, e._datasize)
    assert n == 5 and float(back.value) == float(e.value)
    assert back._minibatchsize == e._minibatchsize
    for p in params:
        assert torch.equal(back._acc_stats[p], e._acc_stats[p])
    same, n2 = all_reduce_elbo(e, model, 9)          # no process group: identity
    assert same is e and n2 == 9


def test_sharding_balances_frames():
    lengths = [400, 390, 380, 50, 40, 30, 20, 10, 395, 5]
    shards = [shard_utterances(lengths, 4, r) for r in range(4)]
    assert sorted(sum(shards, [])) == list(range(len(lengths)))
    loads = [sum(lengths[u] for u in s) for s in shards]
    assert max(loads) - min(loads) <= 60                # split -n l/N would give 1170 vs 35
