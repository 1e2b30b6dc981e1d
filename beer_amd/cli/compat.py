"""Load pickles written by the reference implementation (`beer.*` classes)
into the `beer_amd` classes.  Works because the host layer keeps the
reference's attribute names (buffers `mean`, `scale`, ...; `means_precisions`,
`categorical`, `graph`, `start_pdf`, ...): an `nn.Module` pickle is its
`__dict__`, so only the class lookup has to be redirected."""

import importlib
import pickle

__all__ = ['load', 'loads', 'load_npz', 'Unpickler', 'reference_aliases']

_SEARCH = ('beer_amd.models', 'beer_amd.dists', 'beer_amd.graph', 'beer_amd.inference',
           'beer_amd.inference.objectives', 'beer_amd.cli.dataset', 'beer_amd.dists.expfam',
           'beer_amd.dists.families')


class Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if module == 'beer' or module.startswith('beer.'):
            for target in _SEARCH:
                mod = importlib.import_module(target)
                if hasattr(mod, name):
                    return getattr(mod, name)
            raise pickle.UnpicklingError(
                f'{module}.{name} has no counterpart in beer_amd (out of scope)')
        return super().find_class(module, name)


def load(fileobj):
    'pickle.load that maps `beer.*` classes to `beer_amd.*`.'
    return Unpickler(fileobj).load()


def loads(data):
    import io
    return load(io.BytesIO(data))


def load_npz(path):
    """{name: array} of an `.npz` archive whose object arrays may pickle `beer.*`
    classes (the `alis.npz` of `beer hmm mkaligraph`): every member is read with
    numpy's own header parser and, where it holds objects, unpickled by `Unpickler`
    above -- no entry of `sys.modules` is touched, so a real `beer` package
    imported elsewhere in the process (another thread, a CPU baseline) is never
    shadowed."""
    import zipfile
    import numpy as np
    from numpy.lib import format as npf
    out = {}
    with zipfile.ZipFile(path) as archive:
        for member in archive.namelist():
            name = member[:-4] if member.endswith('.npy') else member
            with archive.open(member) as fp:
                version = npf.read_magic(fp)
                if version == (1, 0):
                    shape, fortran, dtype = npf.read_array_header_1_0(fp)
                elif version == (2, 0):
                    shape, fortran, dtype = npf.read_array_header_2_0(fp)
                else:
                    raise ValueError(f'{path}:{member}: .npy format version {version}')
                if dtype.hasobject:
                    out[name] = Unpickler(fp).load()
                else:
                    data = np.frombuffer(fp.read(), dtype=dtype)
                    out[name] = data.reshape(shape[::-1]).T if fortran else data.reshape(shape)
    return out


class _Alias:
    'Stand-in for a `beer.*` module: attribute lookups resolve in beer_amd.'

    def __init__(self, name):
        self.__name__ = name

    def __getattr__(self, name):
        for target in _SEARCH:
            mod = importlib.import_module(target)
            if hasattr(mod, name):
                return getattr(mod, name)
        raise AttributeError(name)


_REFERENCE_MODULES = (
    'beer', 'beer.graph', 'beer.models', 'beer.dists', 'beer.inference',
    'beer.inference.objectives', 'beer.cli', 'beer.cli.dataset',
    'beer.models.hmm', 'beer.models.phoneloop', 'beer.models.mixture',
    'beer.models.mixtureset', 'beer.models.normalset', 'beer.models.normal',
    'beer.models.categorical', 'beer.models.categoricalset', 'beer.models.modelset',
    'beer.models.parameters', 'beer.models.basemodel', 'beer.dists.normalwishart',
    'beer.dists.normalgamma', 'beer.dists.isonormalgamma', 'beer.dists.dirichlet',
    'beer.dists.gamma', 'beer.dists.basedist')


class reference_aliases:
    '''(Kept for callers that unpickle through third-party code; `load` / `load_npz`
    need no aliasing and are what the command line uses.)
    Context manager: while active, pickles that name `beer.*` classes (e.g.
    the object arrays inside an `alis.npz` written by the reference) resolve to
    beer_amd classes -- always, whether or not a `beer` package is installed next
    to beer_amd: the product path never hands its objects to the reference.  The
    module table is restored on exit (an installed `beer` is untouched outside).'''

    def __enter__(self):
        import sys
        self._saved = {}
        for name in _REFERENCE_MODULES:
            self._saved[name] = sys.modules.get(name)
            sys.modules[name] = _Alias(name)
        return self

    def __exit__(self, *exc):
        import sys
        for name, old in self._saved.items():
            if old is None:
                sys.modules.pop(name, None)
            else:
                sys.modules[name] = old
        return False
