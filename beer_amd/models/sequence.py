"""HMM and PhoneLoop.

API mirror of beer/models/hmm.py:13-121 and beer/models/phoneloop.py:13-101.
The per-utterance methods below drive the same ragged-batch kernels as the
batched accumulator (beer_amd/inference/batch.py) with a batch of one.
"""

import torch

from .. import _hip, hmm_kernels as hk, kernels
from ..stats import reference_layout_enabled
from .basemodel import DiscreteLatentModel
from .gaussians import NormalSet
from .modelset import DynamicallyOrderedModelSet
from .weights import Categorical, SBCategorical

__all__ = ['HMM', 'PhoneLoop']


class HMM(DiscreteLatentModel):
    'Hidden Markov Model with fixed transition probabilities.'

    @classmethod
    def create(cls, graph, modelset):
        return cls(graph, modelset)

    def __init__(self, graph, modelset):
        super().__init__(DynamicallyOrderedModelSet(modelset))
        self.graph = graph

    # -- helpers ---------------------------------------------------------------
    def _emissions(self):
        return self.modelset.original_modelset

    def _pc_llhs(self, stats, inference_graph):
        order = inference_graph.pdf_id_mapping
        return self.modelset.expected_log_likelihood(stats, order)

    def _batch_of_one(self, graph, n_frames, dtype):
        return hk.HmmBatch([graph], [0], [n_frames], dtype)

    def _inference(self, pc_llhs, inference_graph, viterbi=False, state_path=None,
                   trans_posteriors=False):
        '''State posteriors (+ summed transition posteriors) from per-state
        log-likelihoods [T, S]; hmm.py:40-62.'''
        pc = _hip.on_device(pc_llhs)
        batch = self._batch_of_one(inference_graph, len(pc), pc.dtype)
        flat = pc.reshape(-1)
        if viterbi or state_path is not None:
            path = hk.viterbi(batch, flat) if state_path is None else state_path
            gamma, xi, g0 = hk.path_posteriors(batch, path, want_xi=trans_posteriors)
        else:
            gamma, xi, g0, _, _ = hk.forward_backward(batch, flat, want_xi=trans_posteriors,
                                                      dense_xi=True)
        gamma = gamma.view(len(pc), -1)
        return ((gamma, xi) if trans_posteriors else gamma), None

    # -- Model interface ---------------------------------------------------------
    def mean_field_factorization(self):
        return self.modelset.mean_field_factorization()

    def sufficient_statistics(self, data):
        return self.modelset.sufficient_statistics(data)

    def expected_log_likelihood(self, stats, inference_graph=None, viterbi=False,
                                state_path=None, scale=1., utt_lengths=None):
        '''sum_s gamma_ts * scale * l_ts per frame (hmm.py:73-92).  Without an
        inference graph the model's own graph is used and the transition
        posteriors are kept (PhoneLoop needs them).  `utt_lengths` (not in
        the reference) treats the frames as that many consecutive utterances,
        each decoded with the same graph, in one ragged batch.'''
        trans_posts = inference_graph is None
        graph = self.graph if inference_graph is None else inference_graph
        dense = kernels.is_dense(stats)
        emissions = self._emissions()
        # (no gradient through the posteriors: detached statistics, hmm.py:81-87)
        pc_all = emissions.expected_log_likelihood(stats.detach())
        self.modelset.cache['order'] = graph.pdf_id_mapping
        T, S_total = pc_all.shape
        if utt_lengths is None:
            batch = self._batch_of_one(graph, T, pc_all.dtype)
        else:
            lengths = [int(n) for n in utt_lengths]
            if sum(lengths) != T:
                raise ValueError('utt_lengths do not add up to the number of frames')
            batch = hk.HmmBatch([graph], [0] * len(lengths), lengths, pc_all.dtype)
        flow = g0 = None
        # a ragged batch (`utt_lengths`, not in the reference) on graphs the one-wave kernel
        # takes: gather + forward-backward + scatter in ONE launch, as `accumulate_elbo` runs
        # an HMM shard (the per-state posteriors [T, n_states] are not kept then -- a phone
        # loop counts from the first-frame posteriors and hub flows the kernel sums)
        fused = utt_lengths is not None and not viterbi and state_path is None and \
            hk.fused_ok(batch)
        need_counts = trans_posts and hasattr(self, 'start_pdf')
        if fused and need_counts:
            fused = getattr(getattr(batch.dgraphs[0], 'lowdeg', None), 'n_hubs', 0) >= 1
        if fused:
            # (the per-frame value sum_s gamma l comes out of the same launch)
            exp_llh = torch.empty(T, dtype=pc_all.dtype, device=pc_all.device)
            state_resps, g0, flow = hk.posteriors_fused(batch, pc_all, scale,
                                                        want_counts=need_counts, frame_llh=exp_llh)
            self.cache.pop('resps', None)
            if trans_posts:
                self.cache['trans_resps'] = None
                self.cache['hub_flow'] = flow
                self.cache['first_resps'] = g0
            self.cache['scaled_pdf_resps'] = state_resps
            self.cache['scale'] = scale
            return self._with_gradient(stats, exp_llh, state_resps, emissions, dense)
        pc_llhs = hk.gather(batch, pc_all, scale)
        if viterbi or state_path is not None:
            path = hk.viterbi(batch, pc_llhs) if state_path is None else state_path
            gamma, xi, g0 = hk.path_posteriors(batch, path, want_xi=trans_posts)
        else:
            per_frame = trans_posts and reference_layout_enabled() and utt_lengths is None
            # (per frame: the general kernel -- log-space forward values, hub arcs in the matrix)
            gamma, xi, g0, _, flow = hk.forward_backward(batch, pc_llhs,
                                                         want_xi=trans_posts and not per_frame,
                                                         dense_xi=per_frame)
            if per_frame:
                # the reference's [T-1, S, S] tensor, hub arcs included: no separate flows
                xi = hk.trans_posteriors_dense(batch, pc_llhs, gamma, graph.trans_log_probs)
                flow = None
        state_resps, exp_llh = hk.scatter(batch, pc_llhs, gamma, S_total, scale)
        self.cache['resps'] = gamma.view(T, -1)
        if trans_posts:
            # summed over time: [S, S]; transitions through a declared hub (phone
            # loop) are summed over their sources in 'hub_flow' [S] instead
            self.cache['trans_resps'] = xi
            self.cache['hub_flow'] = flow
            # posteriors of the FIRST frame, summed over the utterances of a ragged batch
            # (`utt_lengths`): what a phone loop counts besides the flows (phoneloop.py:88-95,
            # once per utterance in the reference's loop)
            self.cache['first_resps'] = g0
        self.cache['scaled_pdf_resps'] = state_resps
        self.cache['scale'] = scale
        return self._with_gradient(stats, exp_llh, state_resps, emissions, dense)

    @staticmethod
    def _with_gradient(stats, exp_llh, state_resps, emissions, dense):
        'The value with its gradient w.r.t. differentiable statistics / frames (a VAE\'s prior).'
        if dense and isinstance(emissions, NormalSet):
            # statistics-in (prior of a VAE): d exp_llh / d stats through
            # sum_s gamma_ts * scale * l_ts with detached posteriors (hmm.py:81-87)
            exp_llh = kernels.attach_stats_grad(
                stats, exp_llh, state_resps, emissions.means_precisions.natural_form())
        elif isinstance(emissions, NormalSet) and kernels.has_source(stats):
            # the same for statistics that are phi(z_t) of differentiable frames (a VAE with
            # one sample per frame): the frame kernels above, the gradient w.r.t. the frames
            exp_llh = kernels.attach_frame_grad(
                stats, exp_llh, state_resps, emissions.means_precisions.natural_form())
        return exp_llh

    def accumulate(self, stats, parent_msg=None):
        # scale * resps scattered back to pdf ids was produced with the E-step.
        return {**self._emissions().accumulate(stats, self.cache['scaled_pdf_resps'])}

    # -- DiscreteLatentModel interface ------------------------------------------------
    def decode(self, data, inference_graph=None, scale=1.):
        'Viterbi path mapped to pdf ids, LongTensor on the host (hmm.py:105-114).'
        graph = self.graph if inference_graph is None else inference_graph
        stats = self.sufficient_statistics(data)
        pc_all = self._emissions().expected_log_likelihood(stats)
        batch = self._batch_of_one(graph, len(stats), pc_all.dtype)
        pc_llhs = hk.gather(batch, pc_all, scale)
        return hk.viterbi(batch, pc_llhs, map_pdf=True).cpu()

    def posteriors(self, data, inference_graph=None, scale=1.0):
        'State posteriors; `scale` multiplies the statistics (hmm.py:116-121).'
        graph = self.graph if inference_graph is None else inference_graph
        stats = self.modelset.sufficient_statistics(data) * scale
        pc_all = self._emissions().expected_log_likelihood(stats)
        batch = self._batch_of_one(graph, len(stats), pc_all.dtype)
        gamma = hk.forward_backward(batch, hk.gather(batch, pc_all, 1.))[0]
        return gamma.view(len(stats), -1)


class PhoneLoop(HMM):
    'Phone-loop HMM with a prior over the phone (unigram) weights.'

    @classmethod
    def create(cls, graph, start_pdf, end_pdf, modelset, categorical=None,
               prior_strength=1.0):
        tensor = modelset.mean_field_factorization()[0][0].prior._tensors()[0]
        if categorical is None:
            weights = torch.ones(len(start_pdf), dtype=tensor.dtype, device=tensor.device)
            weights /= len(start_pdf)
            categorical = Categorical.create(weights, prior_strength)
        return cls(graph, modelset, start_pdf, end_pdf, categorical)

    def __init__(self, graph, modelset, start_pdf, end_pdf, categorical):
        super().__init__(graph, modelset)
        self.start_pdf = start_pdf
        self.end_pdf = end_pdf
        self.categorical = categorical
        param = self.categorical.mean_field_factorization()[0][0]
        param.register_callback(self._on_weights_update)
        self._on_weights_update()

    def _index_tensors(self, device):
        '''(end states, start states) of the phones as device index tensors,
        built once per device: a host -> device copy from pageable memory
        behind queued kernels blocks the host for tens of ms (it showed up as
        a 50 ms stall every time the phone weights were updated).'''
        end_idxs, start_idxs = list(self.end_pdf.values()), list(self.start_pdf.values())
        memo = self.__dict__.get('_idx_memo')
        if memo is None or memo[0] != device or memo[1] != (end_idxs, start_idxs):
            memo = (device, (end_idxs, start_idxs), torch.as_tensor(end_idxs, device=device),
                    torch.as_tensor(start_idxs, device=device))
            self.__dict__['_idx_memo'] = memo
        return memo[2], memo[3]

    def _on_weights_update(self):
        '''Rewrite the phone-exit transitions with E[ln w] (phoneloop.py:53-65).
        Host-side callback over P x P entries; the CSR copy on the device is
        rebuilt at the next inference (CompiledGraph.device_graph).'''
        trans = self.graph.trans_log_probs
        log_weights = self.categorical.log_weights().to(dtype=trans.dtype,
                                                        device=trans.device)
        start_idxs = list(self.start_pdf.values())
        end_idxs = list(self.end_pdf.values())
        # all phones at once (the reference loops over them; same elementwise ops)
        ends, starts = self._index_tensors(trans.device)
        residuals = (1 - trans[ends, ends].exp()).log()
        if len(set(end_idxs)) == len(end_idxs):
            trans[ends[:, None], starts[None, :]] = residuals[:, None] + log_weights[None, :]
        else:                                    # repeated end states: last write wins
            for i, end_idx in enumerate(end_idxs):
                trans[end_idx, start_idxs] = residuals[i] + log_weights
        self.graph.weights_rewritten()
        # the block just written is rank one: tell the graph, so that
        # forward-backward can treat the eliminated pivot state as a hub
        if len(set(end_idxs)) == len(end_idxs) and len(set(start_idxs)) == len(start_idxs):
            self.graph.set_hub(end_idxs, residuals, start_idxs, log_weights)

    # the callback is index arithmetic on device tensors when the graph lives on the GPU
    # (no host copy, no synchronisation): the update of the weights' group may then be
    # captured as a HIP graph (parameters.py: register_callback)
    def _weights_update_capturable(self):
        if not self.graph.trans_log_probs.is_cuda:
            return False
        # (the index tensors are made NOW: their host -> device copy cannot be recorded)
        self._index_tensors(self.graph.trans_log_probs.device)
        return True

    _on_weights_update.device_only = _weights_update_capturable

    def mean_field_factorization(self):
        from .mixtures import _merge_groups
        return _merge_groups(self.modelset.mean_field_factorization(),
                             self.categorical.mean_field_factorization())

    def phone_counts(self, xi_sum, gamma0, hub_flow=None):
        '''sum_t xi_t[ends, starts] summed over ends + gamma_0[starts] (88-95).
        `xi_sum` may be None when the graph routes every end -> start arc through
        its hub: `hub_flow` then holds the whole sum.'''
        ends, starts = self._index_tensors(gamma0.device)
        counts = gamma0[starts].to(torch.float64)
        if xi_sum is not None and xi_sum.dim() == 3:          # per frame (reference layout)
            xi_sum = xi_sum.sum(dim=0)
        if xi_sum is not None:
            counts = counts + xi_sum[:, starts][ends, :].sum(dim=0)
        if hub_flow is not None:
            counts = counts + hub_flow[starts]
        return counts

    def accumulate(self, stats, parent_msg=None):
        retval = super().accumulate(stats, parent_msg)
        wparam = self.categorical.mean_field_factorization()[0][0]
        ref = wparam.stats
        if 'trans_resps' in self.cache:
            first = self.cache.get('first_resps')
            counts = self.phone_counts(self.cache['trans_resps'],
                                       self.cache['resps'][0] if first is None else first,
                                       self.cache.get('hub_flow'))
            counts = counts.to(dtype=ref.dtype, device=ref.device)
            resps_stats = self.categorical.sufficient_statistics(counts.view(1, -1))
            retval.update(self.categorical.accumulate(resps_stats))
        else:
            # trained with forced alignments: the phone weights get no counts
            fake = torch.zeros(len(self.start_pdf), dtype=ref.dtype, device=ref.device)
            retval.update(self.categorical.accumulate(fake[None, :]))
        return retval
