"""HMM topology: the Python graph builder and the compiled inference graph.

API mirror of beer/graph.py (Graph 61-240, CompiledGraph 243-344).  Building
and compiling a graph is host-side bookkeeping and stays in Python; inference
(`posteriors`, `best_path`) runs the HIP kernels of beer_amd/csrc/hmm.hip on a
CSR image of the transition matrix that `CompiledGraph` keeps on the device.
"""

import ctypes
from collections import OrderedDict

import numpy as np
import torch

from . import _hip

__all__ = ['Graph', 'CompiledGraph', 'GraphSet', 'SparseGraph', 'compile_alignments']


class State:
    'Node of a graph (plain attributes: pickles of the reference restore into it).'

    def __init__(self, id, pdf_id):
        self.id, self.pdf_id = id, pdf_id

    def __repr__(self):
        return f'State(id={self.id}, pdf_id={self.pdf_id})'


class Arc:
    'Weighted arc; identity (hash / equality) is the (start, end) pair.'

    def __init__(self, start, end, weight=1.0):
        self.start, self.end, self.weight = start, end, weight

    def __hash__(self):
        return hash((self.start, self.end))

    def __eq__(self, other):
        return (self.start, self.end) == (other.start, other.end)

    def __repr__(self):
        return f'Arc(start={self.start}, end={self.end}, weight={self.weight})'


class Graph:
    'Directed graph of emitting (pdf_id set) and non-emitting states.'

    def __init__(self):
        self._state_count = 0
        self._states = OrderedDict()
        self._arcs = set()
        self.symbols = {}
        self.start_state = None
        self.end_state = None

    def states(self):
        return self._states.keys()

    def state_from_id(self, state_id):
        return self._states[state_id]

    def arcs(self, state_id=None, incoming=False):
        'All arcs, or the outgoing / incoming arcs of `state_id`.'
        for arc in self._arcs:
            if state_id is None or (arc.end if incoming else arc.start) == state_id:
                yield arc

    def add_state(self, pdf_id=None):
        state_id = self._state_count
        self._state_count += 1
        self._states[state_id] = State(state_id, pdf_id)
        return state_id

    def add_arc(self, start, end, weight=1.0):
        arc = Arc(start, end, weight)
        self._arcs.add(arc)
        return arc

    def normalize(self):
        'Make the outgoing weights of every state sum to one.'
        for state_id in self.states():
            out = list(self.arcs(state_id))
            total = 0.
            for arc in out:
                total += arc.weight
            for arc in out:
                arc.weight /= total

    def replace_state(self, old_state_id, graph):
        'Substitute a whole (unit) graph for one state.'
        new_ids = {}
        for state_id in graph.states():
            new_ids[state_id] = self.add_state(pdf_id=graph._states[state_id].pdf_id)
        for arc in graph.arcs():
            self.add_arc(new_ids[arc.start], new_ids[arc.end], arc.weight)
        stale, fresh = [], []
        for arc in self.arcs(old_state_id):
            stale.append(arc)
            fresh.append((new_ids[graph.end_state], arc.end, arc.weight))
        for arc in self.arcs(old_state_id, incoming=True):
            stale.append(arc)
            fresh.append((arc.start, new_ids[graph.start_state], arc.weight))
        for start, end, weight in fresh:
            self.add_arc(start, end, weight)
        for arc in stale:
            self._arcs.remove(arc)
        del self._states[old_state_id]

    def _walk(self, start_state, init_weight, incoming):
        'Follow non-emitting states until emitting ones; yields (state, weight).'
        frontier = [(arc, init_weight) for arc in self.arcs(start_state, incoming=incoming)]
        visited = {start_state}
        while frontier:
            arc, weight = frontier.pop()
            nxt = arc.start if incoming else arc.end
            if self._states[nxt].pdf_id is not None:
                yield nxt, weight * arc.weight
            elif nxt not in visited:
                frontier += [(a, arc.weight * weight)
                             for a in self.arcs(nxt, incoming=incoming)]
                visited.add(nxt)

    def find_next_pdf_ids(self, start_state, init_weight=1.0):
        return self._walk(start_state, init_weight, incoming=False)

    def find_previous_pdf_ids(self, start_state, init_weight=1.0):
        return self._walk(start_state, init_weight, incoming=True)

    def _topology(self):
        'Flat arrays of the graph for the native compiler.'
        ids = list(self._states)
        local = {sid: i for i, sid in enumerate(ids)}
        pdf = np.asarray([-1 if self._states[s].pdf_id is None else int(self._states[s].pdf_id)
                          for s in ids], dtype=np.int32)
        arcs = list(self._arcs)
        src = np.asarray([local[a.start] for a in arcs], dtype=np.int32)
        dst = np.asarray([local[a.end] for a in arcs], dtype=np.int32)
        w = np.asarray([a.weight for a in arcs], dtype=np.float64)
        return pdf, src, dst, w, local[self.start_state], local[self.end_state]

    def compile(self):
        '''Remove the non-emitting states: initial / final / transition
        probabilities between emitting states (graph.py:185-240), rows
        renormalised without changing the self-loop probability.  Runs in the
        native compiler (`beer_graph_compile`, O(states + arcs)).'''
        pdf, src, dst, w, start, end = self._topology()
        handle = ctypes.c_void_p()
        _hip.call_host('beer_graph_compile', len(pdf), _np_ptr(pdf), len(src), _np_ptr(src),
                       _np_ptr(dst), _np_ptr(w), start, end, ctypes.byref(handle))
        return GraphSet(handle)[0].to_dense()


def _np_ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a.size else None


class GraphSet:
    '''Compiled graphs held by the native library (`beer_graphset`): CSR
    arcs with float32 probabilities, as the reference's tables before
    `.log()`.  Indexing yields `SparseGraph` views; `device_image(dtype)` lays
    ALL graphs out in one blob and copies it to the GPU once.'''

    def __init__(self, handle):
        self._handle = handle
        n = ctypes.c_int64()
        _hip.call_host('beer_graphset_sizes', handle, ctypes.byref(n), None, None)
        self.n = n.value
        self.state_off = np.zeros(self.n + 1, dtype=np.int64)
        self.arc_off = np.zeros(self.n + 1, dtype=np.int64)
        _hip.call_host('beer_graphset_sizes', handle, ctypes.byref(n), _np_ptr(self.state_off),
                       _np_ptr(self.arc_off))
        S, A = int(self.state_off[-1]), int(self.arc_off[-1])
        self.init = np.zeros(S, dtype=np.float32)
        self.final = np.zeros(S, dtype=np.float32)
        self.pdf_ids = np.zeros(S, dtype=np.int32)
        self.arc_src = np.zeros(A, dtype=np.int32)
        self.arc_dst = np.zeros(A, dtype=np.int32)
        self.arc_prob = np.zeros(A, dtype=np.float32)
        _hip.call_host('beer_graphset_export', handle, _np_ptr(self.init), _np_ptr(self.final),
                       _np_ptr(self.pdf_ids), _np_ptr(self.arc_src), _np_ptr(self.arc_dst),
                       _np_ptr(self.arc_prob))
        self._images = {}
        # largest in / out degree over all graphs (beer_batch.max_degree)
        gidx = np.repeat(np.arange(self.n), np.diff(self.arc_off))
        deg = 1
        if A:
            base = self.state_off[gidx]
            deg = max(int(np.bincount(base + self.arc_src, minlength=S).max()),
                      int(np.bincount(base + self.arc_dst, minlength=S).max()), 1)
        self.max_degree = deg

    def __del__(self):
        handle, self._handle = getattr(self, '_handle', None), None
        if handle:
            try:
                _hip.call_host('beer_graphset_free', handle)
            except Exception:                               # interpreter shutdown
                pass

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if not 0 <= i < self.n:
            raise IndexError(i)
        return SparseGraph(self, i)

    def __iter__(self):
        return (SparseGraph(self, i) for i in range(self.n))

    def device_image(self, dtype):
        '(blob on the GPU, ctypes array of beer_graph descriptors), built once per dtype.'
        memo = self._images.get(dtype)
        if memo is None:
            dev = _hip.require_device()
            code = _hip.dtype_code(dtype)
            nbytes = ctypes.c_size_t()
            _hip.call_host('beer_graphset_image_bytes', self._handle, code, ctypes.byref(nbytes))
            host = torch.empty(max(16, nbytes.value), dtype=torch.uint8, pin_memory=True)
            blob = torch.empty(max(16, nbytes.value), dtype=torch.uint8, device=dev)
            structs = (_hip.Graph * max(1, self.n))()
            _hip.call_host('beer_graphset_image', self._handle, code,
                           ctypes.c_void_p(host.data_ptr()), blob.data_ptr(), structs)
            blob.copy_(host, non_blocking=True)
            self._host_image = host                  # keep the pinned source until the copy ran
            memo = self._images[dtype] = (blob, structs)
        return memo


class _ArenaDeviceGraph:
    'Descriptor of one graph of a GraphSet image (what HmmBatch reads).'

    def __init__(self, struct, keep):
        self.struct, self._keep = struct, keep
        self.n_states, self.n_arcs = struct.n_states, struct.n_arcs
        self.n_in_seg, self.n_out_seg = struct.n_in_seg, struct.n_out_seg
        self.lowdeg = True if struct.lowdeg else None
        if self.lowdeg:
            self.lowdeg_info = (keep[0].max_degree, 0, 0)        # alignment chains: no hub


class SparseGraph:
    '''One compiled graph of a GraphSet.  Stands in for `CompiledGraph`
    wherever an inference graph is expected (`inference_graph=`,
    `accumulate_elbo(..., inference_graphs=)`); `to_dense()` gives the
    reference's dense-matrix object (e.g. to pickle it).'''

    def __init__(self, owner, i):
        self._set, self._i = owner, i
        s0, s1 = owner.state_off[i], owner.state_off[i + 1]
        self.n_states = int(s1 - s0)
        self.pdf_id_mapping = owner.pdf_ids[s0:s1]

    def _arcs(self):
        o = self._set
        a0, a1 = o.arc_off[self._i], o.arc_off[self._i + 1]
        return o.arc_src[a0:a1], o.arc_dst[a0:a1], o.arc_prob[a0:a1]

    @property
    def init_log_probs(self):
        o = self._set
        return torch.from_numpy(o.init[o.state_off[self._i]:o.state_off[self._i + 1]]).log()

    @property
    def final_log_probs(self):
        o = self._set
        return torch.from_numpy(o.final[o.state_off[self._i]:o.state_off[self._i + 1]]).log()

    @property
    def trans_log_probs(self):
        src, dst, prob = self._arcs()
        dense = torch.zeros(self.n_states, self.n_states)
        dense[torch.from_numpy(src).long(), torch.from_numpy(dst).long()] = torch.from_numpy(prob)
        return dense.log()

    def to_dense(self):
        return CompiledGraph(self.init_log_probs, self.final_log_probs, self.trans_log_probs,
                             [int(i) for i in self.pdf_id_mapping])

    def device_graph(self, dtype):
        blob, structs = self._set.device_image(dtype)
        return _ArenaDeviceGraph(structs[self._i], (self._set, blob))

    def posteriors(self, llhs, trans_posteriors=False):
        return CompiledGraph.posteriors(self, llhs, trans_posteriors)

    def best_path(self, llhs):
        return CompiledGraph.best_path(self, llhs)


def compile_alignments(sequences, units):
    '''Alignment graphs of many transcriptions in one native call
    (`beer_aligraphs_compile`): `sequences` is a list of unit-name sequences,
    `units` maps a unit name to its `Graph` (what `beer hmm mkphones` writes).
    Returns a `GraphSet`; graph u equals
    `create_graph_from_seq(sequences[u], units)` of the reference
    (mkaligraph.py:18-39).'''
    names = list(units)
    uid = {name: i for i, name in enumerate(names)}
    state_off, arc_off = [0], [0]
    pdfs, starts, ends, srcs, dsts, ws = [], [], [], [], [], []
    for name in names:
        pdf, src, dst, w, start, end = units[name]._topology()
        pdfs.append(pdf)
        srcs.append(src)
        dsts.append(dst)
        ws.append(w)
        starts.append(start)
        ends.append(end)
        state_off.append(state_off[-1] + len(pdf))
        arc_off.append(arc_off[-1] + len(src))
    i32 = lambda v: np.ascontiguousarray(v, dtype=np.int32)              # noqa: E731
    cat = lambda parts, dt: np.ascontiguousarray(np.concatenate(parts), dtype=dt)  # noqa: E731
    seq_off = np.zeros(len(sequences) + 1, dtype=np.int64)
    seq_off[1:] = np.cumsum([len(s) for s in sequences])
    flat = i32([uid[u] for seq in sequences for u in seq])
    args = [i32(state_off), cat(pdfs, np.int32), i32(starts), i32(ends), i32(arc_off),
            cat(srcs, np.int32), cat(dsts, np.int32), cat(ws, np.float64)]
    handle = ctypes.c_void_p()
    _hip.call_host('beer_aligraphs_compile', len(names), *[_np_ptr(a) for a in args],
                   len(sequences), _np_ptr(seq_off), _np_ptr(flat), ctypes.byref(handle))
    return GraphSet(handle)


class DeviceGraph:
    'CSR image of a CompiledGraph in device memory + its beer_graph struct.'

    def __init__(self, init, final, trans, device, dtype, hubs=()):
        trans_h = trans.detach().to('cpu', torch.float64)
        S = trans_h.shape[0]
        finite = trans_h > -float('inf')
        # by destination, sources ascending (Viterbi's first-index tie-break)
        dst, src = torch.nonzero(finite.t(), as_tuple=True)
        in_ptr = torch.zeros(S + 1, dtype=torch.int32)
        in_ptr[1:] = torch.cumsum(torch.bincount(dst, minlength=S), 0).to(torch.int32)
        in_w = trans.detach().to('cpu')[src, dst]
        # by source, destinations ascending
        src2, dst2 = torch.nonzero(finite, as_tuple=True)
        out_ptr = torch.zeros(S + 1, dtype=torch.int32)
        out_ptr[1:] = torch.cumsum(torch.bincount(src2, minlength=S), 0).to(torch.int32)
        out_w = trans.detach().to('cpu')[src2, dst2]
        self.n_states, self.n_arcs = S, int(src.numel())
        in_seg, in_row_seg = self._segments(in_ptr)
        out_seg, out_row_seg = self._segments(out_ptr)
        self.n_in_seg, self.n_out_seg = len(in_seg) - 1, len(out_seg) - 1
        i32 = lambda t: t.to(torch.int32)                       # noqa: E731
        self.bufs = _hip.upload(dict(
            init=init.detach().to('cpu', dtype), final=final.detach().to('cpu', dtype),
            in_ptr=in_ptr, in_src=i32(src), in_dst=i32(dst), in_w=in_w.to(dtype),
            in_seg=in_seg, in_row_seg=in_row_seg,
            out_ptr=out_ptr, out_dst=i32(dst2), out_src=i32(src2), out_w=out_w.to(dtype),
            out_seg=out_seg, out_row_seg=out_row_seg), device)
        b = self.bufs
        p = lambda name: b[name].data_ptr()                     # noqa: E731
        self.lowdeg = self._lowdeg(trans.detach().to('cpu'), finite, hubs, device, dtype)
        lowdeg_ptr = 0
        if self.lowdeg is not None:
            b['lowdeg_struct'] = _hip.upload(
                {'s': torch.frombuffer(bytearray(bytes(self.lowdeg)), dtype=torch.uint8)},
                device)['s']
            lowdeg_ptr = b['lowdeg_struct'].data_ptr()
        self.struct = _hip.Graph(S, self.n_arcs, self.n_in_seg, self.n_out_seg,
                                 p('init'), p('final'),
                                 p('in_ptr'), p('in_src'), p('in_dst'), p('in_w'),
                                 p('in_seg'), p('in_row_seg'),
                                 p('out_ptr'), p('out_dst'), p('out_src'), p('out_w'),
                                 p('out_seg'), p('out_row_seg'), lowdeg_ptr)

    def _lowdeg(self, trans_h, finite, hubs, device, dtype):
        '''Factorised image (beer_graph_lowdeg): the hub blocks -- verified to be
        rank one in the log domain -- are removed from the CSR; kept only if
        every remaining row has at most SEG arcs.'''
        S = trans_h.shape[0]
        if len(hubs) > _hip.MAX_HUBS:
            return None
        keep = finite.clone()
        src_id = torch.full((S,), -1, dtype=torch.int32)
        dst_id = torch.full((S,), -1, dtype=torch.int32)
        src_w = torch.zeros(S, dtype=trans_h.dtype)
        dst_w = torch.zeros(S, dtype=trans_h.dtype)
        src_ptr, src_list, dst_ptr, dst_list = [0], [], [0], []
        for h, (src, sw, dst, dw) in enumerate(hubs):
            src, dst = torch.as_tensor(src).cpu().long(), torch.as_tensor(dst).cpu().long()
            # The dense matrix is authoritative: factor the block itself
            # (A[e,s] = A[e,0] + A[0,s] - A[0,0] for a rank-one log block) and
            # use the caller's weights only as a hint that the block exists.
            block = trans_h[src][:, dst].to(torch.float64)
            if not bool(torch.isfinite(block).all()):
                return None
            sw = block[:, 0].clone()
            dw = block[0, :] - block[0, 0]
            model = sw[:, None] + dw[None, :]
            # as exact as the image's arithmetic: a float32-rounded table is rank one
            # only to 1e-7, which a float64 image must not silently inherit (it then
            # keeps the dense block: the general kernel)
            tol = 1e-12 if dtype == torch.float64 else 1e-6
            ok = torch.allclose(block, model, rtol=0., atol=tol * float(block.abs().max())) and \
                bool((src_id[src] < 0).all()) and bool((dst_id[dst] < 0).all())
            sw, dw = sw.to(trans_h.dtype), dw.to(trans_h.dtype)
            if not ok:
                return None
            keep[src[:, None], dst[None, :]] = False
            src_id[src], dst_id[dst] = h, h
            src_w[src], dst_w[dst] = sw, dw
            src_list += src.tolist()
            src_ptr.append(len(src_list))
            dst_list += dst.tolist()
            dst_ptr.append(len(dst_list))
        if keep.sum(0).max() > _hip.SEG or keep.sum(1).max() > _hip.SEG:
            return None
        dst_i, src_i = torch.nonzero(keep.t(), as_tuple=True)
        in_ptr = torch.zeros(S + 1, dtype=torch.int32)
        in_ptr[1:] = torch.cumsum(torch.bincount(dst_i, minlength=S), 0).to(torch.int32)
        src_o, dst_o = torch.nonzero(keep, as_tuple=True)
        out_ptr = torch.zeros(S + 1, dtype=torch.int32)
        out_ptr[1:] = torch.cumsum(torch.bincount(src_o, minlength=S), 0).to(torch.int32)
        i32 = lambda v: torch.as_tensor(v, dtype=torch.int32)                  # noqa: E731
        b = self.bufs
        ld = _hip.upload(dict(
            ld_in_ptr=in_ptr, ld_in_src=i32(src_i), ld_in_w=trans_h[src_i, dst_i].to(dtype),
            ld_in_dst=i32(dst_i), ld_out_src=i32(src_o),
            ld_out_ptr=out_ptr, ld_out_dst=i32(dst_o), ld_out_w=trans_h[src_o, dst_o].to(dtype),
            ld_src_id=src_id, ld_src_w=src_w.to(dtype), ld_dst_id=dst_id, ld_dst_w=dst_w.to(dtype),
            ld_src_ptr=i32(src_ptr), ld_src_list=i32(src_list if src_list else [0]),
            ld_dst_ptr=i32(dst_ptr), ld_dst_list=i32(dst_list if dst_list else [0])), device)
        ld['_ld_blob'] = ld.pop('_blob')
        b.update(ld)
        self._hub_ptr = (list(src_ptr), list(dst_ptr))
        # (largest degree of the image's CSR, hubs, largest hub side): what the batch
        # descriptor tells the kernels (beer_batch.max_degree ...)
        members = [src_ptr[h + 1] - src_ptr[h] for h in range(len(hubs))] + \
            [dst_ptr[h + 1] - dst_ptr[h] for h in range(len(hubs))] + [0]
        self.lowdeg_info = (max(1, int(keep.sum(0).max()), int(keep.sum(1).max())), len(hubs),
                            max(members))
        q = lambda name: b[name].data_ptr()                      # noqa: E731
        return _hip.GraphLowDeg(
            int(src_i.numel()), len(hubs), q('ld_in_ptr'), q('ld_in_src'), q('ld_in_w'),
            q('ld_out_ptr'), q('ld_out_dst'), q('ld_out_w'), q('ld_src_id'), q('ld_src_w'),
            q('ld_dst_id'), q('ld_dst_w'), q('ld_src_ptr'), q('ld_src_list'), q('ld_dst_ptr'),
            q('ld_dst_list'))

    def refresh(self, trans):
        '''New transition log-probabilities on the SAME sparsity pattern (the
        phone-loop weights were rewritten): every weight array of the image
        is re-gathered from the dense matrix by device index tensors -- no
        host round trip, the descriptors keep their addresses.'''
        b = self.bufs
        dtype = b['in_w'].dtype
        t = trans.detach()
        idx = lambda name: b[name].long()                        # noqa: E731
        b['in_w'].copy_(t[idx('in_src'), idx('in_dst')].to(dtype))
        b['out_w'].copy_(t[idx('out_src'), idx('out_dst')].to(dtype))
        if self.lowdeg is not None:
            if b['ld_in_w'].numel():
                b['ld_in_w'].copy_(t[idx('ld_in_src'), idx('ld_in_dst')].to(dtype))
                b['ld_out_w'].copy_(t[idx('ld_out_src'), idx('ld_out_dst')].to(dtype))
            for h in range(self.lowdeg.n_hubs):
                memo = self.__dict__.setdefault('_hub_index', {})
                if h not in memo:
                    src = b['ld_src_list'][self._hub_ptr[0][h]:self._hub_ptr[0][h + 1]].long()
                    dst = b['ld_dst_list'][self._hub_ptr[1][h]:self._hub_ptr[1][h + 1]].long()
                    # one-element index tensors, not 0-dim ones: torch turns a 0-dim
                    # index into a Python int with .item() -- a device synchronisation
                    # per VB iteration that kept the host from queueing ahead
                    memo[h] = (src, dst, src[:1], dst[:1])
                src, dst, src0, dst0 = memo[h]
                # A[e, s] = A[e, 0] + (A[0, s] - A[0, 0]), as when the image was built
                b['ld_src_w'][src] = t[src, dst0].to(dtype)
                b['ld_dst_w'][dst] = (t[src0, dst].double() - t[src0, dst0].double()).to(dtype)

    @staticmethod
    def _segments(ptr):
        'Cut every CSR row into runs of <= SEG arcs: (arc offsets, row -> seg range).'
        ptr = ptr.tolist()
        seg, row_seg = [], [0]
        for r in range(len(ptr) - 1):
            for beg in range(ptr[r], ptr[r + 1], _hip.SEG):
                seg.append(beg)
            row_seg.append(len(seg))
        seg.append(ptr[-1])
        return (torch.tensor(seg, dtype=torch.int32), torch.tensor(row_seg, dtype=torch.int32))


class CompiledGraph(torch.nn.Module):
    'Inference graph of an HMM: initial, final and transition log-probabilities.'

    def __init__(self, init_log_probs, final_log_probs, trans_log_probs, pdf_id_mapping=None):
        super().__init__()
        self.register_buffer('init_log_probs', init_log_probs)
        self.register_buffer('final_log_probs', final_log_probs)
        self.register_buffer('trans_log_probs', trans_log_probs)
        self.pdf_id_mapping = pdf_id_mapping
        self.hubs = []

    def set_hub(self, src_states, src_log_w, dst_states, dst_log_w):
        '''Declare that trans_log_probs[src, dst] == src_log_w[:, None] +
        dst_log_w[None, :] (the P x P block a phone loop's eliminated pivot
        state leaves behind).  Purely an acceleration hint: the dense matrix
        stays authoritative and the hint is verified before use.'''
        old = self.__dict__.get('hubs') or []       # (absent on graphs unpickled from the reference)
        same = len(old) == 1 and old[0][0] == list(src_states) and old[0][2] == list(dst_states)
        self.hubs = [(list(src_states), src_log_w.detach().clone(), list(dst_states),
                      dst_log_w.detach().clone())]
        if not same:
            self.__dict__.pop('_device_memo', None)

    def weights_rewritten(self):
        '''Tell the graph that `trans_log_probs` was rewritten in place on the
        same sparsity pattern (finite entries stay finite): its device image
        is refreshed on the device instead of being rebuilt -- right here when
        the matrix lives on the GPU (index kernels on the current stream: part of
        a captured M-step when one is being recorded), at the next inference
        otherwise.'''
        memo = self.__dict__.get('_device_memo')
        tensors = (self.init_log_probs, self.final_log_probs, self.trans_log_probs)
        if memo is not None and self.trans_log_probs.is_cuda and \
                all(a is b for a, b in zip(memo[1], tensors)) and \
                memo[2][:2] == tuple(t._version for t in tensors[:2]):
            memo[3].refresh(self.trans_log_probs)
            self.__dict__['_device_memo'] = (memo[0], tensors,
                                             tuple(t._version for t in tensors), memo[3])
            return
        self.__dict__['_weights_rewritten'] = True

    def __repr__(self):
        return '<CompiledGraph>'

    @property
    def n_states(self):
        return len(self.trans_log_probs)

    def __getstate__(self):
        state = self.__dict__.copy()
        state.pop('_device_memo', None)
        state.pop('_weights_rewritten', None)
        state.pop('_batch_cache', None)
        return state

    def device_graph(self, dtype):
        '''CSR copy on the GPU in `dtype`, rebuilt when a buffer is replaced
        or written in place (PhoneLoop rewrites trans_log_probs after every
        update, phoneloop.py:53-65).'''
        tensors = (self.init_log_probs, self.final_log_probs, self.trans_log_probs)
        sig = tuple(t._version for t in tensors)
        memo = self.__dict__.get('_device_memo')
        if memo is not None and memo[0] == dtype and all(a is b for a, b in zip(memo[1], tensors)):
            if memo[2] == sig:
                return memo[3]
            if self.__dict__.pop('_weights_rewritten', False) and memo[2][:2] == sig[:2] and \
                    self.trans_log_probs.device.type == 'cuda':
                memo[3].refresh(self.trans_log_probs)
                self.__dict__['_device_memo'] = (dtype, tensors, sig, memo[3])
                return memo[3]
        self.__dict__.pop('_weights_rewritten', None)
        dg = DeviceGraph(*tensors, device=_hip.require_device(), dtype=dtype,
                         hubs=self.__dict__.get('hubs', ()))
        self.__dict__['_device_memo'] = (dtype, tensors, sig, dg)
        return dg

    def posteriors(self, llhs, trans_posteriors=False):
        '''State posteriors [N, K] (and, summed over time instead of the
        reference's [N-1, K, K] tensor, transition posteriors [K, K]) from
        per-frame, per-state log-likelihoods.  Returns (posteriors,
        mean per-frame log-normaliser) like graph.py:289-326.'''
        from .hmm_kernels import HmmBatch, forward_backward, trans_posteriors_dense
        from .stats import reference_layout_enabled
        batch = HmmBatch([self], [0], [len(llhs)], llhs.dtype, with_pdf_ids=False)
        llhs_d = _hip.on_device(llhs)
        per_frame = trans_posteriors and reference_layout_enabled()
        gamma, xi_sum, _, lognorm, _ = forward_backward(
            batch, llhs_d, want_xi=trans_posteriors and not per_frame, want_lognorm=True,
            dense_xi=True)
        if per_frame:
            # the reference's [N-1, K, K] tensor (beer_amd.reference_layout)
            xi = trans_posteriors_dense(batch, llhs_d, gamma, self.trans_log_probs)
            return (gamma.view(len(llhs), -1), xi), lognorm[0]
        gamma = gamma.view(len(llhs), -1)
        if trans_posteriors:
            return (gamma, xi_sum.to(llhs.dtype)), lognorm[0]
        return gamma, lognorm[0]

    def best_path(self, llhs):
        'Viterbi state path, int64 [N] (graph.py:329-344).'
        from .hmm_kernels import HmmBatch, viterbi
        batch = HmmBatch([self], [0], [len(llhs)], llhs.dtype, with_pdf_ids=False)
        return viterbi(batch, _hip.on_device(llhs), map_pdf=False)
