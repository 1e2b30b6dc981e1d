"""ctypes binding of the C ABI declared in include/beer_hip.h.

torch is used here for what the design allows it to do: own device memory
(`tensor.data_ptr()`), name the current HIP stream and move bytes.  Every
numerical step of the hot path is a call into libbeer_hip.so; there is NO CPU
fallback -- without the library or without a GPU the calls raise.
"""

import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# BEER_HIP_LIB: another build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get('BEER_HIP_LIB') or os.path.join(_HERE, 'csrc', 'libbeer_hip.so')

F32, F64, I16 = 0, 1, 2
FULL, DIAG, ISO = 0, 1, 2
SEG = 8                     # BEER_SEG of include/beer_hip.h
MAX_HUBS = 4                # kMaxHubs of csrc/hmm.hip
COV_CODE = {'full': FULL, 'diagonal': DIAG, 'isotropic': ISO}
EINVAL = -100000


class HipUnavailable(RuntimeError):
    'Raised when the HIP extension or the GPU is missing (no CPU fallback).'


class HipError(RuntimeError):
    'A call into libbeer_hip.so failed; `rc` is its status (-(hipError_t) for launch errors).'
    rc = None


class HipInvalid(HipError):
    '''BEER_EINVAL: the library refused the arguments (a shape this entry point does not
    take) before launching anything.  The only failure a caller may answer by trying
    another entry point; launch errors are plain `HipError`s and must propagate.'''
    rc = EINVAL


_lib = None


def lib():
    'The loaded shared library (loaded once; raises if it was not built).'
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HipUnavailable(
                f'{LIB_PATH} is missing: build it with `python -m beer_amd.build` '
                '(beer_amd has no CPU fallback)')
        _lib = ctypes.CDLL(LIB_PATH)
        _declare(_lib)
        _options_from_env(_lib)
    return _lib


# BEER_OPT_* of include/beer_hip.h and the environment variables that preset them
OPTIONS = {'ax_max_frames': (0, 'BEER_AX_MAXFRAMES'), 'accf_rounds': (1, 'BEER_ACCF_ROUNDS'),
           'k1_wide': (2, 'BEER_K1_WIDE'), 'accfi_waves': (3, 'BEER_ACCFI_WAVES'),
           'lnfi': (4, 'BEER_LNFI'), 'fb_log': (5, 'BEER_FB_LOG'),
           'k1_lds': (6, 'BEER_K1_LDS')}


def _options_from_env(l):
    for name, (code, env) in OPTIONS.items():
        text = os.environ.get(env)
        if text is None:
            continue
        try:
            value = int(text)
        except ValueError:
            raise ValueError(f'{env}={text!r}: expected an integer') from None
        if l.beer_hip_set_option(code, value) != 0:
            raise ValueError(f'{env}={value}: outside the range of option {name!r} '
                             '(include/beer_hip.h)')


def set_option(name, value):
    '''Set a launch-tuning option of the library (`OPTIONS`; include/beer_hip.h
    BEER_OPT_*); returns the previous value.  Out-of-range values raise.'''
    code = OPTIONS[name][0]
    old = lib().beer_hip_get_option(code)
    if lib().beer_hip_set_option(code, int(value)) != 0:
        raise ValueError(f'option {name!r}: value {value} out of range')
    return old


def get_option(name):
    return lib().beer_hip_get_option(OPTIONS[name][0])


c_p, c_i, c_l, c_d = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double
c_z = ctypes.c_size_t


class Graph(ctypes.Structure):
    'beer_graph of include/beer_hip.h.'
    _fields_ = [('n_states', ctypes.c_int32), ('n_arcs', ctypes.c_int32),
                ('n_in_seg', ctypes.c_int32), ('n_out_seg', ctypes.c_int32),
                ('init', c_p), ('final', c_p),
                ('in_ptr', c_p), ('in_src', c_p), ('in_dst', c_p), ('in_w', c_p),
                ('in_seg', c_p), ('in_row_seg', c_p),
                ('out_ptr', c_p), ('out_dst', c_p), ('out_src', c_p), ('out_w', c_p),
                ('out_seg', c_p), ('out_row_seg', c_p), ('lowdeg', c_p)]


class GraphLowDeg(ctypes.Structure):
    'beer_graph_lowdeg of include/beer_hip.h.'
    _fields_ = [('n_arcs', ctypes.c_int32), ('n_hubs', ctypes.c_int32),
                ('in_ptr', c_p), ('in_src', c_p), ('in_w', c_p),
                ('out_ptr', c_p), ('out_dst', c_p), ('out_w', c_p),
                ('hub_src_id', c_p), ('hub_src_w', c_p), ('hub_dst_id', c_p),
                ('hub_dst_w', c_p), ('src_ptr', c_p), ('src_list', c_p),
                ('dst_ptr', c_p), ('dst_list', c_p)]


class Batch(ctypes.Structure):
    'beer_batch of include/beer_hip.h.'
    _fields_ = [('nutt', ctypes.c_int32), ('max_states', ctypes.c_int32),
                ('max_arcs', ctypes.c_int32), ('max_segs', ctypes.c_int32),
                ('all_lowdeg', ctypes.c_int32), ('n_graphs', ctypes.c_int32),
                ('frame_off', c_p), ('llh_off', c_p), ('graph_id', c_p),
                ('graphs', c_p), ('pdf_off', c_p), ('pdf_ids', c_p),
                ('max_degree', ctypes.c_int32), ('max_hubs', ctypes.c_int32),
                ('max_hub_members', ctypes.c_int32), ('reserved', ctypes.c_int32),
                ('order', c_p)]


class FeaConf(ctypes.Structure):
    'beer_feaconf of include/beer_hip.h.'
    _fields_ = [('flen', ctypes.c_int32), ('fstep', ctypes.c_int32),
                ('fft_len', ctypes.c_int32), ('mode', ctypes.c_int32),
                ('nfilters', ctypes.c_int32), ('apply_log', ctypes.c_int32),
                ('n_dct', ctypes.c_int32), ('add_energy', ctypes.c_int32),
                ('preemph', ctypes.c_double), ('log_offset', ctypes.c_double),
                ('norm', ctypes.c_double),
                ('window', c_p), ('filters', c_p), ('filt_lo', c_p), ('filt_hi', c_p),
                ('dct', c_p), ('lifter', c_p)]


# name -> argument ctypes (return type is always int)
_four = [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]          # dtype,K,D,4 in,out,stream
_from = [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p]          # dtype,K,D,eta,4 out,stream
_dir = [c_i, c_i, c_i, c_p, c_p, c_p]
_gam = [c_i, c_i, c_p, c_p, c_p, c_p]
SIGNATURES = {
    'beer_hip_version': [],
    'beer_hip_device_count': [],
    'beer_hip_set_option': [c_i, c_i],
    'beer_hip_get_option': [c_i],
    'beer_mixtureset_packed_supported': [c_i, c_i, c_i, c_i],
    'beer_nw_expected_stats': _four, 'beer_nw_log_norm': _four, 'beer_nw_natural': _four,
    'beer_nw_from_natural': _from,
    'beer_nw_expected_stats_log_norm': [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    'beer_nw_update': [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    'beer_ng_expected_stats': _four, 'beer_ng_log_norm': _four, 'beer_ng_natural': _four,
    'beer_ng_from_natural': _from,
    'beer_ing_expected_stats': _four, 'beer_ing_log_norm': _four, 'beer_ing_natural': _four,
    'beer_ing_from_natural': _from,
    'beer_dirichlet_expected_stats': _dir, 'beer_dirichlet_log_norm': _dir,
    'beer_dirichlet_natural': _dir, 'beer_dirichlet_from_natural': _dir,
    'beer_dirichlet_log_weights': _dir,
    'beer_sb_transform_stats': [c_i, c_i, c_p, c_p, c_p, c_p],
    'beer_sb_log_weights': [c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    'beer_gamma_expected_stats': _gam, 'beer_gamma_log_norm': _gam,
    'beer_gamma_natural': _gam,
    'beer_gamma_from_natural': [c_i, c_i, c_p, c_p, c_p, c_p],
    'beer_kl_div': [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    'beer_natural_grad_step': [c_i, c_l, c_p, c_p, c_p, c_d, c_p, c_p],
    'beer_suffstats_expand': [c_i, c_i, c_l, c_i, c_p, c_p, c_p],
    'beer_mixtureset_estep': [c_i, c_i, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_d,
                              c_p, c_p, c_p, c_p, c_p, c_z, c_p],
    'beer_normal_accumulate': [c_i, c_i, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_z,
                               c_p],
    'beer_mixture_estep_packed': [c_i, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_z,
                                  c_p],
    'beer_normal_accumulate_packed': [c_i, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_z, c_p],
    'beer_mixtureset_estep_packed': [c_i, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                     c_z, c_p],
    'beer_mixtureset_accumulate_packed': [c_i, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_z,
                                          c_p],
    'beer_unpack_resps': [c_l, c_i, c_p, c_p, c_p],
    'beer_mixtureset_accumulate_fused': [c_i, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p,
                                         c_p, c_p, c_z, c_p],
    'beer_frame_image': [c_i, c_l, c_i, c_p, c_p, c_z, c_p],
    'beer_mixtureset_lognorm_image': [c_i, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                      c_z, c_p],
    'beer_pack_resps': [c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    'beer_weights_from_acc': [c_i, c_i, c_i, c_p, c_p, c_p],
    'beer_hmm_gather': [c_i, c_p, c_i, c_p, c_d, c_p, c_p],
    'beer_hmm_forward_backward': [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    'beer_hmm_posteriors_fused': [c_i, c_p, c_i, c_p, c_d, c_p, c_p, c_p, c_i, c_p, c_p, c_p,
                                  c_p, c_p],
    'beer_hmm_fb_log_count': [c_p, c_p, c_p, c_p],
    'beer_hmm_viterbi': [c_i, c_p, c_p, c_p, c_p, c_i, c_p],
    'beer_hmm_trans_posteriors': [c_i, c_l, c_i, c_p, c_p, c_p, c_p, c_p, c_p],
    'beer_hmm_path_posteriors': [c_i, c_p, c_p, c_p, c_p, c_p, c_p],
    'beer_hmm_scatter': [c_i, c_p, c_i, c_p, c_p, c_d, c_p, c_p, c_p, c_p],
    'beer_segment_sum': [c_i, ctypes.c_int32, c_p, c_p, c_p, c_p],
    'beer_dense_llh': [c_i, c_l, c_i, c_i, c_p, c_p, c_d, c_p, c_p],
    'beer_dense_llh_backward': [c_i, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    'beer_dense_accumulate': [c_i, c_l, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    'beer_rowdot': [c_i, c_l, c_i, c_p, c_p, c_p, c_p],
    'beer_softmax_groups': [c_i, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p],
    'beer_features_signal_mean': [c_i, ctypes.c_int32, c_p, c_p, c_p, c_p],
    'beer_features_extract': [c_i, ctypes.c_int32, c_p, c_p, c_l, c_p, c_p, c_p, c_p,
                              ctypes.c_int32, c_p],
    'beer_features_deltas': [ctypes.c_int32, c_p, c_l, ctypes.c_int32, ctypes.c_int32,
                             ctypes.c_int32, c_p, c_p, c_p],
    'beer_features_cmn': [ctypes.c_int32, c_p, ctypes.c_int32, ctypes.c_int32, c_p, c_p],
    'beer_copy_pinned': [c_p, c_p, c_z, c_p],
    'beer_clock_probe': [c_p, c_i, c_p],
    'beer_suffstats_mean': [c_i, c_i, c_l, c_i, c_i, c_p, c_p, c_p],
    'beer_suffstats_backward': [c_i, c_i, c_l, c_i, c_i, c_p, c_p, c_p, c_p],
    'beer_frames_llh_backward': [c_i, c_i, c_l, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_z, c_p],
}


# host-side functions (graph compilation): host pointers, no stream
c_pp = ctypes.POINTER(ctypes.c_void_p)
HOST_SIGNATURES = {
    'beer_graph_compile': [ctypes.c_int32, c_p, c_l, c_p, c_p, c_p, ctypes.c_int32,
                           ctypes.c_int32, c_pp],
    'beer_aligraphs_compile': [ctypes.c_int32, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_l,
                               c_p, c_p, c_pp],
    'beer_graphset_free': [c_p],
    'beer_graphset_sizes': [c_p, c_p, c_p, c_p],
    'beer_graphset_export': [c_p, c_p, c_p, c_p, c_p, c_p, c_p],
    'beer_graphset_image_bytes': [c_p, c_i, c_p],
    'beer_graphset_image': [c_p, c_i, c_p, ctypes.c_uint64, c_p],
}

# size queries: return a byte count, take no stream
SIZE_QUERIES = {
    'beer_estep_workspace_bytes': [c_i, c_i, c_i, c_i, c_i],
    'beer_accumulate_workspace_bytes': [c_i, c_i, c_i, c_i, c_i],
    'beer_accumulate_frames_workspace_bytes': [c_i, c_i, c_l, c_i, c_i, c_i],
    'beer_packed_resps_bytes': [c_l, c_i, c_i],
    'beer_accumulate_fused_workspace_bytes': [c_i, c_i, c_i, c_i],
    'beer_frame_image_bytes': [c_i, c_l, c_i],
    'beer_accumulate_packed_workspace_bytes': [c_i, c_l, c_i, c_i],
    'beer_mixtureset_accumulate_packed_workspace_bytes': [c_i, c_l, c_i, c_i, c_i],
    'beer_hmm_fb_scratch_doubles': [c_i, c_p, c_i],
    'beer_frames_llh_backward_workspace_bytes': [c_i, c_i, c_l, c_i, c_i],
}


def _declare(l):
    for name, args in SIGNATURES.items():
        fn = getattr(l, name)
        fn.argtypes = args
        fn.restype = c_i
    for name, args in HOST_SIGNATURES.items():
        fn = getattr(l, name)
        fn.argtypes = args
        fn.restype = c_i
    for name, args in SIZE_QUERIES.items():
        fn = getattr(l, name)
        fn.argtypes = args
        fn.restype = c_z


def dtype_code(dtype, exact=False):
    '''BEER_F32 / BEER_F64 of a torch dtype; `exact` adds BEER_EXACT (float32
    products on the exact fp32 MFMA instead of the bf16x3 arithmetic).'''
    if dtype == torch.float32:
        return F32 | EXACT if exact else F32
    if dtype == torch.float64:
        return F64
    raise TypeError(f'beer_amd kernels take float32 or float64 tensors, got {dtype}')


def require_device():
    'Device to compute on.  Raises when there is no GPU: no CPU fallback.'
    if not torch.cuda.is_available():
        raise HipUnavailable('beer_amd needs an AMD GPU (gfx950); there is no CPU fallback')
    lib()
    return torch.device('cuda', torch.cuda.current_device())


def on_device(t, dtype=None):
    '''Contiguous device view/copy of `t` (H2D copy for host tensors: data
    movement only, the arithmetic always runs in the HIP kernels).'''
    dev = require_device()
    if t.device.type != 'cuda':
        t = t.to(dev)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous()


def ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def call(name, *args):
    'Call an entry point on the current torch stream and check its status.'
    rc = getattr(lib(), name)(*args, stream())
    if rc == EINVAL:
        raise HipInvalid(f'{name} failed: invalid argument')
    if rc != 0:
        err = HipError(f'{name} failed: hipError {-rc}')
        err.rc = rc
        raise err


EXACT = 0x10                # BEER_EXACT of include/beer_hip.h: `dtype | EXACT`
F32_MODES = ('exact', 'bf16x3')
# Host-side policy (the library itself keeps no mode: the arithmetic is an
# argument of every call): initial value from BEER_F32_MODE=exact | bf16x3.
_f32_mode = ['exact' if os.environ.get('BEER_F32_MODE') in ('exact', 'f32') else 'bf16x3']


def set_f32_mode(mode):
    '''How float32 models multiply on the matrix cores: 'exact' (fp32 MFMA,
    bitwise an fmaf chain) or 'bf16x3' (default: every operand exactly as three
    bf16 pieces, the six leading partial products of each multiplication on the
    bf16 MFMA, fp32 accumulation -- fp32's own operands and accumulation, product
    error <= 2^-23, at 2.7x the fp32 pipe's rate).  The mode governs the kernels that
    take FRAMES (E-step, accumulation); the statistics-in products of a VAE's prior
    (`beer_dense_*`: large float32 shapes) are bf16x3 in either mode, and the gradient
    w.r.t. the samples of a one-sample VAE (`beer_frames_llh_backward`) leaves the matrix
    cores for a float64-accumulating kernel in 'exact' mode.'''
    if mode not in F32_MODES:
        raise ValueError(f'f32 mode {mode!r}: expected one of {F32_MODES}')
    _f32_mode[0] = mode


def get_f32_mode():
    return _f32_mode[0]


# fewer frames than this always take the exact fp32 kernels: they cost
# microseconds there, and the bf16x3 path has per-call set-up (the packed
# parameter image, the tile images of the hand-over)
FAST_MIN_FRAMES = 16384
MAX_DIM_F32 = 96              # kMaxDimF32 of csrc/estep_mfma.h: the exact fp32 MFMA kernels
MAX_DIM_FAST = 128            # kMaxDimX: the bf16x3 kernels


def f32_fast_ok(X):
    '''True when float32 frames `X` [T, D] take the bf16x3 matrix path: the mode
    is on and there are enough frames.  No look at the data is needed -- three
    bf16 pieces hold any float32 value exactly -- so the answer costs nothing.'''
    return X.dtype == torch.float32 and get_f32_mode() == 'bf16x3' and \
        X.shape[0] >= FAST_MIN_FRAMES and X.shape[1] <= MAX_DIM_FAST


class exact_f32:
    'Context manager: float32 matrix products on the exact fp32 MFMA inside.'

    def __enter__(self):
        self.old = get_f32_mode()
        set_f32_mode('exact')

    def __exit__(self, *exc):
        set_f32_mode(self.old)


def call_host(name, *args):
    'Call a host-side entry point (no stream, works without a GPU).'
    rc = getattr(lib(), name)(*args)
    if rc != 0:
        raise HipInvalid(f'{name} failed: invalid argument')


# Scratch buffers are filed per (query, shape, device, STREAM): two host threads on two streams
# get buffers of their own; two threads sharing ONE stream would share them, which the stream's
# order makes safe (a kernel that uses a workspace is queued before the next one that does).
_workspaces = {}


def workspace(query, dtype, cov, D, S, G, device):
    '''(tensor, nbytes) scratch for the MFMA implementation of a call, or
    (None, 0) when the shape has none.  One buffer per (query, shape, stream)
    is kept and reused: the kernels leave no state in it.'''
    nbytes = getattr(lib(), query)(dtype_code(dtype), cov, D, S, G)
    if nbytes == 0:
        return None, 0
    key = (query, dtype, cov, D, S, G, device, torch.cuda.current_stream().cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf, nbytes


def frames_workspace(dtype, exact, cov, T, D, S, G, device):
    """(tensor, nbytes) scratch of `beer_normal_accumulate` over T frames (for some shapes it
    holds per-chain partial sums, hence T); one buffer per (shape, stream), grown on demand."""
    nbytes = lib().beer_accumulate_frames_workspace_bytes(dtype_code(dtype, exact), cov, T, D, S, G)
    if nbytes == 0:
        return None, 0
    key = ('frames', dtype, cov, D, S, G, device, torch.cuda.current_stream().cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf, nbytes


def packed_workspace(cov, T, D, K, device, sets=None):
    '''(tensor, nbytes) scratch of `beer_normal_accumulate_packed` (grows with T:
    it holds the transposed frames) or, with `sets = (S, G)`, of
    `beer_mixtureset_accumulate_packed` (... and state posteriors); one buffer per
    (shape, stream), grown on demand.'''
    if sets is None:
        nbytes = lib().beer_accumulate_packed_workspace_bytes(cov, T, D, K)
    else:
        nbytes = lib().beer_mixtureset_accumulate_packed_workspace_bytes(cov, T, D, *sets)
    if nbytes == 0:
        return None, 0
    key = ('packed', cov, D, K, device, torch.cuda.current_stream().cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf, nbytes


_staging = threading.local()    # per host thread: .ring [(pinned buffer, event)], .next


def _staging_buffer(nbytes):
    '''A pinned staging buffer of at least `nbytes` from a small ring that is
    allocated once and grown rarely: `hipHostMalloc` takes tens of ms and
    synchronises the device, so it must not happen per batch.'''
    ring = _staging.__dict__.setdefault('ring', [])
    if not ring:
        for _ in range(4):
            ring.append([torch.empty(1 << 20, dtype=torch.uint8, pin_memory=True),
                         torch.cuda.Event()])
    nxt = _staging.__dict__.get('next', 0)
    slot = ring[nxt % len(ring)]
    _staging.next = nxt + 1
    slot[1].synchronize()                      # the copy that last used it has run
    if slot[0].numel() < nbytes:
        slot[0] = torch.empty(max(nbytes, 2 * slot[0].numel()), dtype=torch.uint8,
                              pin_memory=True)
    return slot[0], slot[1]


def upload(tensors, device):
    '''Copy a dict of small host tensors to the GPU as ONE asynchronous
    transfer from pinned memory; returns {name: device view}.  Dozens of
    pageable `tensor.to(device)` copies each make the host wait for the stream
    (and were seen to stall it for tens of ms inside an iteration).'''
    if torch.cuda.is_current_stream_capturing():
        # a captured copy kernel would re-read the pinned staging slot at every replay, long
        # after the ring has handed it to somebody else
        raise HipInvalid('host -> device upload while a HIP graph is being captured: the '
                         'data of a captured iteration must be resident on the device')
    names, metas, total = [], [], 0
    for name, t in tensors.items():
        t = t.detach().contiguous()
        nbytes = t.numel() * t.element_size()
        names.append(name)
        metas.append((t, total, nbytes))
        total = (total + nbytes + 15) // 16 * 16
    host, done = _staging_buffer(max(total, 16))
    for t, off, nbytes in metas:
        if nbytes:
            host[off:off + nbytes] = t.reshape(-1).view(torch.uint8)
    nbytes = (max(total, 16) + 15) // 16 * 16
    blob = torch.empty(nbytes, dtype=torch.uint8, device=device)
    call('beer_copy_pinned', ptr(blob), ctypes.c_void_p(host.data_ptr()), nbytes)
    done.record()
    out = {}
    for name, (t, off, nbytes) in zip(names, metas):
        out[name] = blob[off:off + nbytes].view(t.dtype).view(t.shape)
    out['_blob'] = blob
    return out


def to_device(host_tensor, device=None):
    'One small host tensor to the GPU through the pinned staging ring (see `upload`).'
    device = device or require_device()
    return upload({'t': host_tensor}, device)['t']


def struct_to_device(obj, device):
    'Copy a ctypes structure (or array of them) to device memory as bytes.'
    buf = torch.frombuffer(bytearray(bytes(obj)), dtype=torch.uint8)
    return buf.to(device)
