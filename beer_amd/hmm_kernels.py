"""Python faces of the HMM entry points of include/beer_hip.h: ragged-batch
descriptors, pdf-id gather / scatter, forward-backward, Viterbi."""

import ctypes
import threading

import numpy as np
import torch

from . import _hip

__all__ = ['HmmBatch', 'gather', 'forward_backward', 'viterbi', 'path_posteriors',
           'scatter', 'gather_columns', 'scatter_columns', 'segment_sum', 'fused_ok',
           'posteriors_fused', 'trans_posteriors_dense', 'counting_log_space']


class HmmBatch:
    '''beer_batch descriptor: `graphs` are the distinct CompiledGraph objects
    of the batch, `graph_ids[u]` the one utterance u uses, `lengths[u]` its
    number of frames.  Packed per-state buffers hold utterance u at element
    offset `llh_off[u]` as a row-major [T_u, S_u] block.'''

    def __init__(self, graphs, graph_ids, lengths, dtype, with_pdf_ids=True, lowdeg=True):
        dev = _hip.require_device()
        self.dtype, self.device = dtype, dev
        self.nutt = len(lengths)
        lengths_t = torch.as_tensor(lengths, dtype=torch.int64)
        gid_t = torch.as_tensor(graph_ids, dtype=torch.int64)
        gset = getattr(graphs[0], '_set', None) if len(graphs) > 1 else None
        if gset is not None and all(getattr(g, '_set', None) is gset for g in graphs):
            # graphs of one natively compiled GraphSet: their descriptors are rows
            # of one array -- slice it with numpy instead of building thousands of
            # Python objects per batch (each batch used to cost a 40 ms gen-2
            # garbage collection at 3000 utterances)
            idx = np.fromiter((g._i for g in graphs), dtype=np.int64, count=len(graphs))
            _, structs = gset.device_image(dtype)
            size = ctypes.sizeof(_hip.Graph)
            raw = np.frombuffer(structs, dtype=np.uint8).reshape(-1, size)[idx]
            head = np.ascontiguousarray(raw[:, :16]).view(np.int32)      # n_states, n_arcs, segs
            has_lowdeg = np.ascontiguousarray(raw[:, size - 8:]).view(np.int64).reshape(-1) != 0
            n_states = head[:, 0].tolist()
            max_arcs = int(head[:, 1].max())
            max_segs = int(head[:, 2:4].max())
            all_lowdeg = bool(has_lowdeg.all())
            graph_bytes = torch.from_numpy(np.ascontiguousarray(raw).reshape(-1))
            so = gset.state_off
            counts = (so[idx + 1] - so[idx]).astype(np.int64)
            pdf_off = np.concatenate([[0], np.cumsum(counts)])
            if with_pdf_ids:
                start = np.repeat(so[idx] - pdf_off[:-1], counts)
                pdf_ids = gset.pdf_ids[start + np.arange(pdf_off[-1])]
            else:
                pdf_ids = (np.arange(pdf_off[-1]) - np.repeat(pdf_off[:-1], counts)).astype(np.int32)
            self.dgraphs = [gset]                               # keeps the image alive
            ld_info = (gset.max_degree, 0, 0)                   # alignment chains: no hub
        else:
            self.dgraphs = [g.device_graph(dtype) for g in graphs]
            n_states = [dg.n_states for dg in self.dgraphs]
            parts = [np.asarray(g.pdf_id_mapping, dtype=np.int32)
                     if (with_pdf_ids and g.pdf_id_mapping is not None)
                     else np.arange(g.n_states, dtype=np.int32) for g in graphs]
            pdf_off = np.concatenate([[0], np.cumsum([len(p) for p in parts])]) if parts else [0]
            pdf_ids = np.concatenate(parts) if parts else np.zeros(0, dtype=np.int32)
            arr = (_hip.Graph * len(graphs))(*[dg.struct for dg in self.dgraphs])
            graph_bytes = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            max_arcs = max([dg.n_arcs for dg in self.dgraphs] + [1])
            max_segs = max([max(dg.n_in_seg, dg.n_out_seg) for dg in self.dgraphs] + [1])
            all_lowdeg = bool(self.dgraphs) and \
                all(getattr(dg, 'lowdeg', None) is not None for dg in self.dgraphs)
            infos = [getattr(dg, 'lowdeg_info', (0, 0, 0)) for dg in self.dgraphs] or [(0, 0, 0)]
            known = all(i[0] > 0 for i in infos)
            ld_info = tuple(max(i[k] for i in infos) if known else 0 for k in range(3))
        states_t = torch.as_tensor(n_states, dtype=torch.int64)[gid_t] \
            if self.nutt else torch.zeros(0, dtype=torch.int64)
        frame_off = torch.zeros(self.nutt + 1, dtype=torch.int64)
        frame_off[1:] = torch.cumsum(lengths_t, 0)
        sizes = lengths_t * states_t
        llh_off = torch.zeros(self.nutt + 1, dtype=torch.int64)
        llh_off[1:] = torch.cumsum(sizes, 0)
        self.n_frames = int(frame_off[-1])
        self.n_elems = int(llh_off[-1])
        self.frame_off_h, self.llh_off_h = frame_off, llh_off
        self.n_states = n_states
        self.graph_ids = list(graph_ids)
        # longest utterance first: the waves of a workgroup of the one-wave kernels then
        # finish together, and a launch ends with its shortest utterances
        order = torch.argsort(lengths_t, descending=True, stable=True).to(torch.int32) \
            if self.nutt else torch.zeros(0, dtype=torch.int32)
        self.bufs = _hip.upload(dict(
            frame_off=frame_off, llh_off=llh_off[:-1], graph_id=gid_t.to(torch.int32),
            order=order,
            graphs=graph_bytes,
            pdf_off=torch.as_tensor(np.asarray(pdf_off), dtype=torch.int32),
            pdf_ids=torch.as_tensor(pdf_ids, dtype=torch.int32)), dev)
        b = self.bufs
        self.struct = _hip.Batch(
            self.nutt, max(n_states) if n_states else 1, max(max_arcs, 1), max(max_segs, 1),
            1 if (lowdeg and all_lowdeg) else 0, len(graphs),
            b['frame_off'].data_ptr(), b['llh_off'].data_ptr(), b['graph_id'].data_ptr(),
            b['graphs'].data_ptr(), b['pdf_off'].data_ptr(), b['pdf_ids'].data_ptr(),
            *(ld_info if all_lowdeg else (0, 0, 0)), 0, b['order'].data_ptr())
        self.shared_graph = len(graphs) == 1
        self._pdf_ids_h, self._pdf_off_h = np.asarray(pdf_ids), np.asarray(pdf_off)
        self._profile = {}

    def pdf_ids_profile(self, S_total):
        '''(some graph repeats a pdf id, every graph's ids are exactly 0 .. S_total-1):
        what decides between atomic adds, plain stores into a zero-filled array and
        plain stores into an uninitialised one when posteriors go back to pdf ids.'''
        hit = self._profile.get(S_total)
        if hit is None:
            ids, off = self._pdf_ids_h, self._pdf_off_h
            n = len(off) - 1
            counts = np.diff(off)
            gidx = np.repeat(np.arange(n), counts)
            key = gidx.astype(np.int64) * (int(ids.max()) + 1 if len(ids) else 1) + ids
            distinct = len(np.unique(key)) == len(key)
            covers = distinct and bool((counts == S_total).all()) and \
                (len(ids) == 0 or int(ids.max()) < S_total)
            hit = self._profile[S_total] = (not distinct, covers)
        return hit

    def ref(self):
        return ctypes.byref(self.struct)


def gather(batch, pc_all, scale=1.):
    'pc_llhs (packed, [n_elems]) = scale * pc_all[frame, pdf_id].'
    pc_all = _hip.on_device(pc_all, batch.dtype)
    out = torch.empty(batch.n_elems, dtype=batch.dtype, device=batch.device)
    _hip.call('beer_hmm_gather', _hip.dtype_code(batch.dtype), batch.ref(), pc_all.shape[1],
              _hip.ptr(pc_all), float(scale), _hip.ptr(out))
    return out


def forward_backward(batch, pc_llhs, want_xi=False, want_lognorm=False, dense_xi=False):
    '''(gamma packed, xi_sum [S,S] fp64 or None, gamma0_sum [S] fp64 or None,
    lognorm_mean [nutt] or None, hub_flow [S] fp64 or None).  xi / gamma0 need
    a batch sharing one graph.  When the graph carries a hub (phone loop) the
    transition posteriors through it come back summed over its sources in
    `hub_flow`; `dense_xi=True` forces the general kernel and a complete
    [S, S] matrix.'''
    dt, dev = batch.dtype, batch.device
    st = batch.struct
    lowdeg_flag = st.all_lowdeg
    gamma = torch.empty(batch.n_elems, dtype=dt, device=dev)
    alpha = torch.empty(batch.n_elems, dtype=torch.float64, device=dev)
    xi = g0 = ln = flow = None
    if want_xi:
        if not batch.shared_graph:
            raise ValueError('transition posteriors need one graph for the whole batch')
        S = batch.n_states[0]
        xi = torch.zeros(S, S, dtype=torch.float64, device=dev)
        g0 = torch.zeros(S, dtype=torch.float64, device=dev)
        flow = torch.zeros(S, dtype=torch.float64, device=dev)
    if want_lognorm:
        ln = torch.empty(batch.nutt, dtype=dt, device=dev)
    try:
        if dense_xi:
            # the general kernel: a complete [S, S] matrix, and log-space forward values in
            # `alpha` (what `trans_posteriors_dense` reads; the one-wave kernel keeps scaled
            # probabilities there).  The descriptor is shared with later calls on the same
            # batch: the flag is put back below.
            st.all_lowdeg = 0
        # hub values per frame, or -- graphs too dense for the arc lists to sit in LDS --
        # the general kernel's per-arc scratch
        n_ws = max(_hip.MAX_HUBS * batch.n_frames,
                   _hip.lib().beer_hmm_fb_scratch_doubles(_hip.dtype_code(dt), batch.ref(),
                                                          int(want_xi)))
        hub_ws = torch.empty(n_ws, dtype=torch.float64, device=dev)
        # which kernel family the C entry point picks (wave_fb_ok of csrc/hmm.hip): the
        # one-wave kernel leaves SCALED PROBABILITIES in `alpha`, the others logarithms
        alpha_is_log = not fused_ok(batch)
        _hip.call('beer_hmm_forward_backward', _hip.dtype_code(dt), batch.ref(),
                  _hip.ptr(pc_llhs), _hip.ptr(alpha), _hip.ptr(hub_ws), _hip.ptr(gamma),
                  _hip.ptr(xi), _hip.ptr(g0), _hip.ptr(flow), _hip.ptr(ln))
    except _hip.HipError as err:
        if 'invalid argument' in str(err) and not st.all_lowdeg:
            raise _hip.HipError(
                f'forward-backward: a graph of the batch has {st.max_states} states; the general '
                "kernel keeps five per-state arrays in the CU's 160 KB of LDS (about 4000 states). "
                'Graphs with at most 8 arcs per state besides a declared hub '
                '(CompiledGraph.set_hub) run at any size') from err
        raise
    finally:
        st.all_lowdeg = lowdeg_flag
    batch.last_alpha = alpha              # (for `trans_posteriors_dense`)
    batch.last_alpha_is_log = alpha_is_log
    if not alpha_is_log:
        counting_log_space.note(batch, hub_ws)
    return gamma, xi, g0, ln, flow


def trans_posteriors_dense(batch, pc_llhs, gamma, trans_log_probs):
    '''Per-frame transition posteriors [T-1, S, S] of a ONE-utterance batch whose
    forward-backward call has just run (`beer_hmm_trans_posteriors`): the
    reference's layout, graph.py:308-323.'''
    if batch.nutt != 1:
        raise ValueError('per-frame transition posteriors: one utterance at a time')
    if not getattr(batch, 'last_alpha_is_log', False):
        raise ValueError('per-frame transition posteriors read log-space forward values: run '
                         'forward_backward(..., dense_xi=True) on this batch first')
    S, T = batch.n_states[0], batch.n_frames
    trans = _hip.on_device(trans_log_probs, batch.dtype)
    xi = torch.zeros(max(T - 1, 0), S, S, dtype=batch.dtype, device=batch.device)
    _hip.call('beer_hmm_trans_posteriors', _hip.dtype_code(batch.dtype), T, S,
              _hip.ptr(batch.last_alpha), _hip.ptr(pc_llhs), _hip.ptr(gamma), _hip.ptr(trans),
              _hip.ptr(xi))
    return xi


class counting_log_space:
    '''Context manager (per host thread): while active, every one-wave forward-backward
    launch adds the number of utterances it ran in LOG SPACE to `.count` (0-dim int64
    device tensor; `beer_hmm_fb_log_count`).  The one-wave kernels work on scaled
    probabilities and hand an utterance over to their log-space twin when a column or a
    frame's normaliser leaves fp64's range or a log-likelihood is NaN (the reference is
    log-space throughout, graph.py:270-326); `.launches` counts the launches seen.'''
    _tls = threading.local()

    def __enter__(self):
        self.count = torch.zeros((), dtype=torch.int64, device=_hip.require_device())
        self.launches = 0
        self._outer = getattr(self._tls, 'active', None)
        self._tls.active = self
        return self

    def __exit__(self, *exc):
        self._tls.active = self._outer
        return False

    @classmethod
    def note(cls, batch, hub_ws):
        self = getattr(cls._tls, 'active', None)
        if self is not None and fused_ok(batch):
            _hip.call('beer_hmm_fb_log_count', batch.ref(), _hip.ptr(hub_ws),
                      _hip.ptr(self.count))
            self.launches += 1


FUSED_MAX_STATES = 256      # kWvMaxStates of csrc/hmm.hip


def fused_ok(batch):
    '''True when `posteriors_fused` takes the batch: low-degree graphs of <= 256
    states with at most one hub of <= 64 members a side (wave_fb_ok of csrc/hmm.hip).'''
    st = batch.struct
    return bool(st.all_lowdeg) and st.max_states <= FUSED_MAX_STATES and \
        1 <= st.max_degree <= _hip.SEG and st.max_hubs <= 1 and st.max_hub_members <= 64


FUSED_ROW_MAX = 512          # kWvRowMax of hmm.hip: pdf ids of a set that fit a wave's LDS row


def posteriors_fused(batch, pc_all, scale=1., want_counts=False, utt_llh=None, frame_llh=None):
    '''Gather + forward-backward + scatter of a shard in one launch
    (`beer_hmm_posteriors_fused`): (state_resps [n_frames, S_total] = scale *
    gamma at the pdf ids, gamma0_sum [S] or None, hub_flow [S] or None);
    `utt_llh` [nutt] fp64 += sum_t sum_s gamma * scale * pc; `frame_llh` [n_frames]
    (the batch's dtype) receives that sum per frame (hmm.py:87).  `want_counts`
    (one graph for the whole batch): the posteriors of the first frame and the
    flows through the graph's hub -- what PhoneLoop counts (phoneloop.py:88-95).'''
    dt, dev = batch.dtype, batch.device
    pc_all = _hip.on_device(pc_all, dt)
    S_total = pc_all.shape[1]
    repeats, covers = batch.pdf_ids_profile(S_total)
    # how the posteriors go back to pdf ids (include/beer_hip.h): whole rows through LDS when a
    # graph repeats ids or leaves some out (alignment graphs) and a row fits; else atomic adds
    # into / plain stores over a zero-filled array; plain stores when the ids are a permutation
    out_mode = 2 if (repeats or not covers) and S_total <= FUSED_ROW_MAX else (1 if repeats else 0)
    make = torch.empty if (out_mode == 2 or (not repeats and covers)) else torch.zeros
    sr = make(batch.n_frames, S_total, dtype=dt, device=dev)
    alpha = torch.empty(batch.n_elems, dtype=torch.float64, device=dev)
    hub_ws = torch.empty(_hip.MAX_HUBS * batch.n_frames, dtype=torch.float64, device=dev)
    g0 = flow = None
    if want_counts:
        if not batch.shared_graph:
            raise ValueError('first-frame posteriors / hub flows need one graph for the batch')
        S = batch.n_states[0]
        g0 = torch.zeros(S, dtype=torch.float64, device=dev)
        flow = torch.zeros(S, dtype=torch.float64, device=dev)
    if frame_llh is not None and (frame_llh.dtype != dt or frame_llh.numel() != batch.n_frames or
                                  not frame_llh.is_contiguous() or frame_llh.device != sr.device):
        raise ValueError('frame_llh: a contiguous [n_frames] tensor of the batch\'s dtype and device')
    _hip.call('beer_hmm_posteriors_fused', _hip.dtype_code(dt), batch.ref(), S_total,
              _hip.ptr(pc_all), float(scale), _hip.ptr(alpha), _hip.ptr(hub_ws), _hip.ptr(sr),
              out_mode, _hip.ptr(g0), _hip.ptr(flow), _hip.ptr(utt_llh),
              _hip.ptr(frame_llh))
    counting_log_space.note(batch, hub_ws)
    return sr, g0, flow


def viterbi(batch, pc_llhs, map_pdf=False):
    'int64 state (or pdf id) path for every frame of the batch.'
    bt = torch.empty(batch.n_elems, dtype=torch.int32, device=batch.device)
    path = torch.empty(batch.n_frames, dtype=torch.int64, device=batch.device)
    _hip.call('beer_hmm_viterbi', _hip.dtype_code(batch.dtype), batch.ref(),
              _hip.ptr(pc_llhs), _hip.ptr(bt), _hip.ptr(path), 1 if map_pdf else 0)
    return path


def path_posteriors(batch, path, want_xi=False):
    dt, dev = batch.dtype, batch.device
    path = _hip.on_device(torch.as_tensor(path)).to(torch.int64).contiguous()
    gamma = torch.empty(batch.n_elems, dtype=dt, device=dev)
    xi = g0 = None
    if want_xi:
        S = batch.n_states[0]
        xi = torch.zeros(S, S, dtype=torch.float64, device=dev)
        g0 = torch.zeros(S, dtype=torch.float64, device=dev)
    _hip.call('beer_hmm_path_posteriors', _hip.dtype_code(dt), batch.ref(), _hip.ptr(path),
              _hip.ptr(gamma), _hip.ptr(xi), _hip.ptr(g0))
    return gamma, xi, g0


def scatter(batch, pc_llhs, gamma, S_total, scale=1., want_resps=True, want_exp_llh=True,
            utt_llh=None):
    '(state_resps [n_frames, S_total] or None, exp_llh [n_frames] or None).'
    dt, dev = batch.dtype, batch.device
    sr = torch.zeros(batch.n_frames, S_total, dtype=dt, device=dev) if want_resps else None
    el = torch.empty(batch.n_frames, dtype=dt, device=dev) if want_exp_llh else None
    _hip.call('beer_hmm_scatter', _hip.dtype_code(dt), batch.ref(), S_total,
              _hip.ptr(pc_llhs), _hip.ptr(gamma), float(scale), _hip.ptr(sr), _hip.ptr(el),
              _hip.ptr(utt_llh))
    return sr, el


def segment_sum(values, frame_off_dev, nutt, out=None):
    'out[u] += sum of values over the frames of utterance u (fp64).'
    values = values.contiguous()
    if out is None:
        out = torch.zeros(nutt, dtype=torch.float64, device=values.device)
    _hip.call('beer_segment_sum', _hip.dtype_code(values.dtype), nutt,
              _hip.ptr(frame_off_dev), _hip.ptr(values), _hip.ptr(out))
    return out


class _Columns:
    'Minimal stand-in for a graph: only n_states / pdf ids matter for gather.'

    def __init__(self, order):
        self.pdf_id_mapping = [int(i) for i in order]
        self.n_states = len(self.pdf_id_mapping)
        self._dg = None

    def device_graph(self, dtype):
        if self._dg is None:
            self._dg = type('DG', (), {})()
            self._dg.n_states, self._dg.n_arcs = self.n_states, 0
            self._dg.n_in_seg = self._dg.n_out_seg = 0
            self._dg.lowdeg = None
            self._dg.struct = _hip.Graph(self.n_states, 0, 0, 0, *([None] * 15))
        return self._dg


def gather_columns(matrix, order):
    'matrix[:, order] on the GPU (modelset.py:140-146).'
    matrix = _hip.on_device(matrix)
    batch = HmmBatch([_Columns(order)], [0], [matrix.shape[0]], matrix.dtype)
    return gather(batch, matrix, 1.).view(matrix.shape[0], len(order))


def scatter_columns(resps, order, n_total):
    'out[:, order[i]] += resps[:, i] (repeated ids add, modelset.py:148-154).'
    resps = _hip.on_device(resps)
    batch = HmmBatch([_Columns(order)], [0], [resps.shape[0]], resps.dtype)
    flat = resps.reshape(-1)
    sr, _ = scatter(batch, flat, flat, n_total, 1., want_exp_llh=False)
    return sr
