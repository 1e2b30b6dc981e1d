"""Variational auto-encoder with an arbitrary prior over the latent space.

API mirror of beer/models/vae.py:28-89.  The encoder / decoder are torch.nn
modules trained by autograd; the prior (Normal, Mixture, HMM ...) is the hot
path.  With one sample per frame (the default) the statistics it receives are
phi(z_t) of the samples: the prior runs its FRAME kernels on them (E-step,
forward-backward, accumulation -- no [T, Q] tensor) and `beer_frames_llh_backward`
carries the gradient back to the encoder.  With several samples per frame it
receives their sample-averaged statistics as a dense [T, Q] tensor
("statistics-in") and runs `beer_dense_llh` / `beer_softmax_groups` /
forward-backward / `beer_dense_accumulate`, with `beer_dense_llh_backward` and
`beer_suffstats_backward` on the way back (`dense_statistics=True`: always).

Quirk Q7 (the DEFAULT, because it is the reference's result): the reference subtracts a
[T] vector from a [T, 1] one (vae.py:84-86) and returns a [T, T] matrix whose sum is T
times the per-frame value -- and T^2 memory.  `VAE(...)` returns T * (llh - kl) per frame:
the same sum, the same gradient scale, the same ELBO as the reference, without the
matrix.  `VAE(..., reference_broadcast=False)` returns the per-frame value the code
evidently meant (a deviation from the reference: ELBO and gradients are T times smaller).
"""

import os

import torch

from .. import _hip, kernels
from ..nnet.linear import Linear
from ..dists.normaldiag import NormalDiagonalCovariance
from .basemodel import Model
from .gaussians import Normal, NormalSet

__all__ = ['VAE']


class MeanLogDiagCov(torch.nn.Module):
    'Normal parameterised by its mean and log-variance (floor 1e-5).'

    def __init__(self, mean, log_diag_cov):
        super().__init__()
        self.mean = mean
        self.log_diag_cov = log_diag_cov

    @property
    def diag_cov(self):
        return 1e-5 + self.log_diag_cov.exp()


class VAE(Model):

    def __init__(self, prior, encoder, decoder, reference_broadcast=True,
                 dense_statistics=None):
        # reference_broadcast=True (default): T x the per-frame value, the reference's own
        # result (its [T, 1] - [T] broadcast, vae.py:84-86); False: the per-frame value
        super().__init__()
        self.prior = prior
        self.encoder = encoder
        self.decoder = decoder
        self.reference_broadcast = reference_broadcast
        # True: the prior always gets dense [T, Q] statistics, also with one sample per frame
        self.dense_statistics = (os.environ.get('BEER_VAE_DENSE') == '1') \
            if dense_statistics is None else bool(dense_statistics)
        self.enc_mean_layer = Linear(encoder.dim_out, decoder.dim_in)
        self.enc_var_layer = Linear(encoder.dim_out, decoder.dim_in)
        self.dec_mean_layer = Linear(decoder.dim_out, encoder.dim_in)
        self.dec_var_layer = Linear(decoder.dim_out, encoder.dim_in)

    def posteriors(self, X):
        'Variational posteriors of the latent variable given the frames.'
        H = self.encoder(X)
        return NormalDiagonalCovariance(
            MeanLogDiagCov(self.enc_mean_layer(H), self.enc_var_layer(H)))

    def pdfs(self, Z):
        'Densities of the frames given the latent variable.'
        H = self.decoder(Z)
        return NormalDiagonalCovariance(
            MeanLogDiagCov(self.dec_mean_layer(H), self.dec_var_layer(H)))

    def _prior_cov_type(self):
        'Covariance type of the Gaussians of the prior (None if it has none).'
        for module in self.prior.modules():
            if isinstance(module, (Normal, NormalSet)):
                param = module.mean_precision if isinstance(module, Normal) \
                    else module.means_precisions
                return param.likelihood_fn.cov_type
        return None

    # -- Model interface -------------------------------------------------------
    def mean_field_factorization(self):
        return self.prior.mean_field_factorization()

    def sufficient_statistics(self, data):
        # The frames go where the NETWORKS are (plain torch.nn, out of the hot path's scope):
        # a notebook that leaves its model on the host (examples/HMM-VAE.ipynb) runs them
        # there; the prior's kernels get the latent samples on the GPU either way.
        return data.to(self.enc_mean_layer.weight.device)

    def expected_log_likelihood(self, data, nsamples=1, llh_weight=1., kl_weight=1.,
                                **kwargs):
        posts = self.posteriors(data)
        T = len(data)

        # local KL divergence by sampling, so that any prior can be plugged in
        samples = posts.sample(nsamples)                          # [T, ns, Dz]
        ent = -posts(posts.sufficient_statistics(samples).mean(dim=1), pdfwise=True)
        flat_nn = samples.reshape(-1, samples.shape[-1])     # where the networks are
        # (the prior's kernels run on the GPU; `.to` is differentiable: gradients flow back)
        flat = flat_nn if flat_nn.is_cuda else flat_nn.to(_hip.require_device())
        cov_type = self._prior_cov_type()
        if cov_type is not None and nsamples == 1 and \
                not getattr(self, 'dense_statistics', False):
            # one sample per frame: the statistics are phi(z_t) -- the prior runs its frame
            # kernels on the samples and differentiates w.r.t. them (kernels.sample_stats)
            prior_stats = kernels.sample_stats(flat, cov_type)
        elif cov_type is not None:
            # sample-averaged statistics [T, Q] in one kernel (no [T*ns, Q] tensor)
            prior_stats = kernels.differentiable_stats(flat, cov_type, nsamples)
        else:
            prior_stats = self.prior.sufficient_statistics(flat)
            prior_stats = prior_stats.reshape(T, nsamples, -1).mean(dim=1)
        # (the accumulation needs the values only: no autograd graph kept in the cache)
        self.cache['prior_stats'] = prior_stats.detach()
        # extra keyword arguments reach the prior (`utt_lengths` of an HMM prior
        # over a batch of utterances); the reference passes none
        xent = -self.prior.expected_log_likelihood(prior_stats, **kwargs).reshape(-1)
        local_kl_div = xent.to(ent.device) - ent

        # expected log-likelihood with the reparameterisation trick
        pdfs = self.pdfs(flat_nn)
        r_data = data[:, None, :].expand(-1, nsamples, -1).reshape(-1, data.shape[-1])
        llh = pdfs(pdfs.sufficient_statistics(r_data), pdfwise=True)
        llh = llh.reshape(T, nsamples).mean(dim=1)

        out = llh_weight * llh - kl_weight * local_kl_div
        return T * out if self.reference_broadcast else out

    def accumulate(self, stats, parent_msg=None):
        return self.prior.accumulate(self.cache['prior_stats'])
