"""Build the HIP kernels + C ABI into beer_amd/csrc/libbeer_hip.so (gfx950).

`python -m beer_amd.build` or `beer_amd.build.build()`.  hipcc cross-compiles
without a GPU; the .so is git-ignored but travels with the tree.
"""

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
ROOT = os.path.dirname(HERE)
LIB = os.path.join(CSRC, 'libbeer_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# -pragma-unroll-threshold: the softmax epilogue of the MFMA E-step must be fully
# unrolled (128 groups) or its accumulators fall out of registers into scratch.
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall',
         '-mllvm', '-pragma-unroll-threshold=262144',
         '-Wno-unused-function', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]


# Per-file flags.  estep_bf16.hip: no SLP vectorisation -- hipcc pairs adjacent float32
# adds / multiplies of the fragment arithmetic into v_pk_* instructions, which beside MFMAs cost
# more than the two plain instructions they replace (MI355X_MICROARCH.md, price of a filler;
# measured: fused accumulation of config 3 10.6 -> 10.3 ms per 3.33 M frames; the gradient
# w.r.t. the samples of config 4 4.21 -> 4.07 ms per 1 M frames).
FILE_FLAGS = {'estep_bf16.hip': ['-fno-slp-vectorize'], 'sample_grad.hip': ['-fno-slp-vectorize']}


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _deps():
    return _sources() + glob.glob(os.path.join(CSRC, '*.h')) + \
        glob.glob(os.path.join(ROOT, 'include', '*.h'))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in _deps())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    for src in _sources():
        obj = src[:-4] + '.o'
        objs.append(obj)
        if not force and os.path.exists(obj) and \
                os.path.getmtime(obj) > max(os.path.getmtime(f) for f in
                                            [src] + glob.glob(os.path.join(CSRC, '*.h')) +
                                            glob.glob(os.path.join(ROOT, 'include', '*.h'))):
            continue
        cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(os.path.basename(src), []) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError(f'hipcc failed on {src}')
        elif verbose and out.strip():
            print(out.decode())
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
