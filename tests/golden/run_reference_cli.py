"""Run the reference's own `beer` command line (build container only: the reference lives
at /root/reference and never travels).  Shims for this image: no `natsort` package,
PyYAML 6 (`yaml.load` needs a Loader), numpy 2 (`np.float` is gone).

    python tests/golden/run_reference_cli.py [-s SEED] <cmd> <subcmd> ...
"""
import re
import runpy
import sys
import types
import warnings

sys.dont_write_bytecode = True
sys.path.insert(0, '/root/reference')
warnings.filterwarnings('ignore')


def _natural(s):
    return [int(t) if t.isdigit() else t for t in re.split(r'(\d+)', s)]


sys.modules['natsort'] = types.SimpleNamespace(
    natsorted=lambda seq, **kw: sorted(seq, key=_natural))
import numpy as np                                      # noqa: E402
if not hasattr(np, 'float'):
    np.float = float
import yaml                                             # noqa: E402
_load = yaml.load
yaml.load = lambda stream, Loader=yaml.SafeLoader: _load(stream, Loader=Loader)
sys.argv[0] = 'beer'
runpy.run_path('/root/reference/beer/cli/beer', run_name='__main__')
