"""RCCL on the one GPU a test box has: a process group of ONE rank over the `nccl` backend
(= RCCL on ROCm), and the data-parallel reduce of beer_amd/distributed.py driven through it --
flatten on the device, `all_reduce(SUM)` of the flat fp64 buffer by RCCL, unflatten with the counts
left on the device, M-step.  A sum over one rank is the identity, so the result must equal the
un-reduced iteration bit for bit; what the test adds to the world-size-2 gloo test
(tests/test_distributed_cpu.py) is that the collective really is RCCL's, on HIP memory, on the
stream the kernels run on.  (Two ranks on one GPU are refused by RCCL: N > 1 needs N GPUs.)"""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

from helpers import load_golden

pytestmark = pytest.mark.gpu

import beer_amd as beer                                             # noqa: E402
from beer_amd.distributed import flatten_elbo, unflatten_elbo        # noqa: E402
from gpu_helpers import build_mixture, npy, params_of, tt            # noqa: E402


def test_all_reduce_of_the_flat_buffer_over_rccl_with_one_rank():
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    device = torch.device('cuda', torch.cuda.current_device())
    dist.init_process_group('nccl', init_method=f'tcp://127.0.0.1:{port}', rank=0, world_size=1,
                            device_id=device)
    try:
        assert dist.get_backend() == 'nccl'
        g = load_golden('g02_gmm_full')
        X = tt(g['X'])
        posts = []
        for reduced in (False, True):
            model = build_mixture(g)
            optim = beer.VBConjugateOptimizer(model.mean_field_factorization(), lrate=1.)
            optim.init_step()
            elbo = beer.evidence_lower_bound(model, X)
            if reduced:
                params = list(model.bayesian_parameters())
                flat = flatten_elbo(elbo, params, 5, device)
                assert flat.is_cuda and flat.dtype == torch.float64
                before = flat.clone()
                dist.all_reduce(flat, op=dist.ReduceOp.SUM)          # RCCL, one rank
                torch.testing.assert_close(flat, before, rtol=0, atol=0)
                # ranks check: an all-reduce of ones counts the ranks the collective spans
                ones = torch.ones(1, dtype=torch.float64, device=device)
                dist.all_reduce(ones)
                assert float(ones) == 1.
                elbo, n = unflatten_elbo(flat, params, elbo._datasize)
                assert isinstance(n, torch.Tensor) and n.is_cuda and int(n) == 5
            elbo.backward()
            optim.step()
            p0, p1 = params_of(model)
            posts.append([npy(getattr(p0.posterior.params, n_)) for n_ in p0.posterior._std_params_def] +
                         [npy(p1.posterior.params.concentrations)])
        for a, b in zip(*posts):
            np.testing.assert_array_equal(a, b)
    finally:
        dist.destroy_process_group()
