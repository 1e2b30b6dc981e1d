"""Lazy sufficient statistics.

The reference materialises phi(X) as a [T, Q] tensor on every call
(Q = D^2+D+2 = 1642 for full covariance at D = 40: 6.6 KB per frame) and
spends 68 % of a GMM step building it.  Here
`Model.sufficient_statistics(X)` returns this handle on the frames instead;
the E-step and accumulation kernels read X directly.  `.dense()` yields the
reference's tensor for callers that really want it.
"""

import torch

from . import _hip

__all__ = ['FrameStats', 'FrameImages', 'reference_layout', 'reference_layout_enabled']

_REFERENCE_LAYOUT = [False]


def reference_layout_enabled():
    return _REFERENCE_LAYOUT[0]


class reference_layout:
    '''Switch (also a context manager) to the reference's own return shapes where
    beer_amd normally keeps something cheaper:

    * `model.sufficient_statistics(X)` returns the dense `[T, Q]` tensor
      (beer/dists/normalwishart.py:30-38) instead of the lazy `FrameStats`;
    * `CompiledGraph.posteriors(llhs, trans_posteriors=True)` and
      `HMM.cache['trans_resps']` hold the transition posteriors per frame,
      `[T-1, S, S]` (beer/graph.py:308-323), instead of their sum over time.

    Meant for small inputs (notebooks, code that inspects these tensors): the
    statistics are 6.6 KB per frame at D = 40, the transition posteriors 8 S^2 bytes
    per frame.  `beer_amd.reference_layout(True)` switches it on for the process,
    `with beer_amd.reference_layout():` for a block.'''

    def __init__(self, enabled=True):
        self._previous = _REFERENCE_LAYOUT[0]
        _REFERENCE_LAYOUT[0] = bool(enabled)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        _REFERENCE_LAYOUT[0] = self._previous
        return False


def _image_bytes(cov_type, T, D):
    return _hip.lib().beer_frame_image_bytes(_hip.COV_CODE[cov_type], T, D)


def _build_image(X, cov_type):
    nbytes = _image_bytes(cov_type, X.shape[0], X.shape[1])
    img = torch.empty(nbytes, dtype=torch.uint8, device=X.device)
    _hip.call('beer_frame_image', _hip.COV_CODE[cov_type], X.shape[0], X.shape[1], _hip.ptr(X),
              _hip.ptr(img), nbytes)
    return img


class FrameImages:
    '''Frame fragment images (include/beer_hip.h: beer_frame_image) of ONE resident tensor
    of float32 frames, owned by whoever owns the frames.

    The fused E-step / accumulation of a mixture set with diagonal or isotropic Gaussians
    reads phi(x) = [x^2, x, 1] of every 32-frame tile as bf16x3 register fragments -- a
    function of the frames only, the same for every component chunk and every VB
    iteration (1056 B per frame at D = 40, 6.6x the frames).  A training loop that walks
    the same shard again and again builds them once:

        X, lengths = beer_amd.pack_utterances(utterances)
        images = beer_amd.FrameImages(X)                  # the caller keeps both alive
        for it in range(n_iter):
            elbo = beer_amd.accumulate_elbo(model, (X, lengths), frame_images=images)

    Without one, `accumulate_elbo` builds the image of each sub-batch for that call only
    (0.26 ms per million frames) and frees it with the sub-batch's scratch.  Nothing is
    cached behind the caller's back: the object holds its images and a reference to `X`
    and nothing else does; `bytes_held` / `frames_bytes` say how much that is; `budget`
    (default min(BEER_FRAME_IMAGE_GB or 64 GB, a quarter of the device)) bounds the images,
    blocks beyond it are rebuilt per call.  Images are filed by the block's offset in `X`;
    an in-place write to `X` (torch's version counter) drops them all -- writes through
    raw pointers or `.data` are the caller's to announce with `clear()`.'''

    def __init__(self, X, budget=None):
        if X.dim() != 2 or X.dtype != torch.float32 or X.device.type != 'cuda':
            raise ValueError('FrameImages: expected float32 frames [T, D] on the GPU')
        self.X = X
        if budget is None:
            import os
            gb = os.environ.get('BEER_FRAME_IMAGE_GB')
            if gb is not None:
                try:
                    gb = float(gb)
                except ValueError:
                    raise ValueError(f'BEER_FRAME_IMAGE_GB={gb!r}: expected a number of '
                                     'gigabytes') from None
            total = torch.cuda.get_device_properties(X.device).total_memory
            budget = int(min(gb * 2 ** 30 if gb is not None else 64 * 2 ** 30, total / 4))
        self.budget = int(budget)
        self._version = X._version
        self._images = {}
        self.builds = self.hits = self.uncached = 0
        self.bytes_held = 0

    @property
    def frames_bytes(self):
        'Bytes of the frames this object keeps alive.'
        return self.X.numel() * self.X.element_size()

    def clear(self):
        self._images.clear()
        self.bytes_held = 0
        self._version = self.X._version

    def covers(self, block):
        'Is `block` a run of whole rows of this object\'s frames?'
        X = self.X
        if block.device != X.device or block.dtype != X.dtype or block.dim() != 2 or \
                block.shape[1] != X.shape[1] or not block.is_contiguous():
            return False
        off = block.data_ptr() - X.data_ptr()
        row = X.shape[1] * X.element_size()
        return X.is_contiguous() and 0 <= off and off % row == 0 and \
            off // row + block.shape[0] <= X.shape[0]

    def get(self, block, cov_type):
        '''The image of `block` (rows of `X`), built on first use; None when the block is
        not part of `X` (the caller builds a temporary one).'''
        if not self.covers(block):
            return None
        if self.X._version != self._version:
            self.clear()
        key = ((block.data_ptr() - self.X.data_ptr()), block.shape[0], cov_type)
        img = self._images.get(key)
        if img is not None:
            self.hits += 1
            return img
        nbytes = _image_bytes(cov_type, block.shape[0], block.shape[1])
        img = _build_image(block, cov_type)
        self.builds += 1
        if self.bytes_held + nbytes <= self.budget:
            self._images[key] = img
            self.bytes_held += nbytes
        else:
            self.uncached += 1
        return img

    def __repr__(self):
        return (f'FrameImages(frames={tuple(self.X.shape)}, images={len(self._images)}, '
                f'bytes_held={self.bytes_held}, builds={self.builds}, hits={self.hits})')


class FrameStats:
    '''phi(X) * scale for a [T, D] block of frames, never formed unless asked.

    Behaves like the tensor the reference returns as far as the hot path uses
    it: `len()`, `.dtype`, `.device`, `.shape`, `stats * scale`.  `images`: the
    caller's `FrameImages` of the tensor these frames are rows of (optional).'''

    def __init__(self, data, cov_type, scale=1.0, images=None):
        if data.dim() != 2:
            raise ValueError('expected a [n_frames, dim] matrix of features')
        self.data = _hip.on_device(data)
        self.cov_type = cov_type
        self.scale = float(scale)
        self.images = images
        self._image = {}                      # this handle's own images, by covariance type
        # the tensor these frames are the values of, when they carry an autograd graph
        # (one sample per frame of a VAE's latent variable): `kernels.sample_stats`
        self.source = None

    def frame_image(self, cov_type=None):
        '''The frame fragment image of these frames for `cov_type` (default: this
        handle's), or None where the kernels take none: from the caller's `FrameImages`
        when there is one, else built once for this handle and freed with it.'''
        import os
        cov_type = cov_type or self.cov_type
        X = self.data
        if os.environ.get('BEER_FRAME_IMAGE', '1') == '0' or X.dtype != torch.float32 or \
                X.shape[0] < _hip.FAST_MIN_FRAMES or _image_bytes(cov_type, *X.shape) == 0:
            return None
        # (filed with the version of the frames it was built from: a handle reused after an
        # in-place write to its frames builds a new image)
        hit = self._image.get(cov_type)
        img = hit[1] if hit is not None and hit[0] == X._version else None
        if img is None and self.images is not None:
            img = self.images.get(X, cov_type)
        if img is None:
            img = _build_image(X, cov_type)
        self._image[cov_type] = (X._version, img)
        return img

    def as_cov(self, cov_type):
        'These frames as statistics of another covariance type (the images are shared).'
        if cov_type == self.cov_type:
            return self
        out = FrameStats(self.data, cov_type, self.scale, self.images)
        out._image = self._image
        out.source = self.source
        return out

    def detach(self):
        'The same frames without the autograd graph of their source (images shared).'
        if self.source is None:
            return self
        out = FrameStats(self.data, self.cov_type, self.scale, self.images)
        out._image = self._image
        return out

    def __len__(self):
        return self.data.shape[0]

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def device(self):
        return self.data.device

    @property
    def dim(self):
        return self.data.shape[1]

    @property
    def shape(self):
        D = self.data.shape[1]
        Q = {'full': D * D + D + 2, 'diagonal': 2 * D + 2, 'isotropic': D + 3}[self.cov_type]
        return torch.Size((self.data.shape[0], Q))

    def __mul__(self, scale):
        out = FrameStats(self.data, self.cov_type, self.scale * float(scale), self.images)
        out._image = self._image
        out.source = self.source
        return out

    __rmul__ = __mul__

    def dense(self):
        'The [T, Q] tensor of the reference (beer_suffstats_expand).'
        T, D = self.data.shape
        out = torch.empty(tuple(self.shape), dtype=self.dtype, device=self.device)
        _hip.call('beer_suffstats_expand', _hip.dtype_code(self.dtype),
                  _hip.COV_CODE[self.cov_type], T, D, _hip.ptr(self.data), _hip.ptr(out))
        if self.scale != 1.0:
            out *= self.scale
        return out

    def __repr__(self):
        return (f'FrameStats(frames={len(self)}, dim={self.dim}, '
                f'cov_type={self.cov_type!r}, scale={self.scale})')
