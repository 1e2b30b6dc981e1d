"""Normal pdf with diagonal covariance whose parameters are network outputs.

API mirror of beer/dists/normaldiag.py:68-190, for the encoder / decoder
densities of a VAE.  Unlike the conjugate families this density is a function
of torch.nn parameters and is differentiated by autograd, so it is written
with plain torch operations (the networks around the
hot path stay torch); the prior over the latent variable, which is the hot
path, goes through the HIP kernels.
"""

import math

import torch

__all__ = ['NormalDiagonalCovariance', 'NormalDiagonalCovarianceStdParams']


class NormalDiagonalCovarianceStdParams(torch.nn.Module):
    'Standard parameters: mean [*, D] and diagonal of the covariance [*, D].'

    def __init__(self, mean, diag_cov):
        super().__init__()
        self.register_buffer('mean', mean)
        self.register_buffer('diag_cov', diag_cov)

    @classmethod
    def from_natural_parameters(cls, natural_params):
        dim = natural_params.shape[-1] // 2
        np1, np2 = natural_params[..., :dim], natural_params[..., dim:2 * dim]
        diag_cov = 1. / (-2 * np2)
        return cls(diag_cov * np1, diag_cov)


def _randn(*shape, **conf):
    'The noise of the reparameterisation trick (a hook for the parity tests).'
    return torch.randn(*shape, **conf)


class NormalDiagonalCovariance(torch.nn.Module):
    '''Set of N Normal pdfs (one per frame).  `params` is any object with
    `mean` and `diag_cov` attributes of shape [N, D].'''

    def __init__(self, params):
        super().__init__()
        self.params = params

    def __len__(self):
        shape = self.params.mean.shape
        return 1 if len(shape) <= 1 else shape[0]

    @property
    def dim(self):
        return self.params.mean.shape[-1]

    def natural_parameters(self):
        'As the reference: [mean / var, 1 / var] (normaldiag.py:168-190).'
        prec = 1. / self.params.diag_cov
        return torch.cat([prec * self.params.mean, prec], dim=-1)

    def sufficient_statistics(self, data):
        return torch.cat([data, -.5 * data ** 2], dim=-1)

    def expected_sufficient_statistics(self):
        mean, var = self.params.mean, self.params.diag_cov
        return torch.cat([mean, -.5 * (var + mean ** 2)], dim=-1)

    def expected_value(self):
        return self.params.mean

    def log_norm(self):
        mean, var = self.params.mean, self.params.diag_cov
        return .5 * (mean ** 2 / var).sum(dim=-1) + .5 * var.log().sum(dim=-1) \
            + .5 * self.dim * math.log(2 * math.pi)

    def forward(self, stats, pdfwise=False):
        '''Log-density of statistics [N, 2D]: pdf n on row n (`pdfwise`) or
        every pdf on every row [N_pdf, N_stats].'''
        mean, var = self.params.mean, self.params.diag_cov
        single = mean.dim() <= 1
        nparams = self.natural_parameters()
        if single:
            mean, var, nparams = mean.view(1, -1), var.view(1, -1), nparams.view(1, -1)
        lnorm = .5 * (var.log().sum(dim=-1) + (mean ** 2 / var).sum(dim=-1))
        base = -.5 * self.dim * math.log(2 * math.pi)
        if pdfwise:
            return torch.sum(nparams * stats, dim=-1) - lnorm + base
        out = nparams @ stats.t() - lnorm[:, None] + base
        return out.reshape(-1) if single else out

    def sample(self, nsamples):
        'mean + sqrt(var) * N(0, I) -> [N, nsamples, D].'
        mean, var = self.params.mean, self.params.diag_cov
        single = mean.dim() == 1
        if single:
            mean, var = mean.view(1, -1), var.view(1, -1)
        noise = _randn(mean.shape[0], nsamples, mean.shape[-1], dtype=mean.dtype,
                       device=mean.device)
        out = mean[:, None, :] + var.sqrt()[:, None, :] * noise
        return out.view(-1, mean.shape[-1]) if single else out
