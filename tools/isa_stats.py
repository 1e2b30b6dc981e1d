#!/usr/bin/env python
"""ISA statistics of the hot bf16x3 kernels (build container; no GPU needed).

    python tools/isa_stats.py [other_source.hip] [extra hipcc flags ...]

Compiles beer_amd/csrc/estep_bf16.hip with -DBEER_KERNEL_PROBE (only the hot kernels are
instantiated) to assembly and prints, per kernel: registers, scratch, and for its
MFMA-heaviest loop the instruction mix (MFMA, VALU, accvgpr moves, LDS, VMEM, scratch,
waits, nops)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'beer_amd', 'csrc', 'estep_bf16.hip')
OUT = '/tmp/isa_probe.s'
if len(sys.argv) > 1 and sys.argv[1].endswith('.hip'):      # another source of the library
    SRC = os.path.join(ROOT, 'beer_amd', 'csrc', sys.argv.pop(1))
cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-mllvm',
       '-pragma-unroll-threshold=262144', '-Wno-unused-function', '-DBEER_KERNEL_PROBE',
       '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.dirname(SRC), '-S',
       '--cuda-device-only', SRC, '-o', OUT] + \
    (['-fno-slp-vectorize'] if SRC.endswith('estep_bf16.hip') else []) + sys.argv[1:]
subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
s = open(OUT).read()
for f in re.split(r'\n\t\.globl\t', s)[1:]:
    name = f.split('\n', 1)[0].strip().split()[0]
    if 'kernel' not in name or name.endswith('.kd'):
        continue
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = dem.replace('beer_mfma::(anonymous namespace)::', '').replace('(anonymous namespace)::', '').replace('void ', '')
    dem = re.sub(r'\(.*', '', dem)
    lines = f.split('\n')
    meta = {k: re.search(r'\.%s:?\s+(\d+)' % k, f) for k in ('vgpr_count', 'agpr_count')}
    g = lambda pat: (re.search(pat, f).group(1) if re.search(pat, f) else '?')
    labels = {m.group(1): i for i, l in enumerate(lines) for m in [re.match(r'^(\.LBB\d+_\d+):', l)] if m}
    best = None
    for i, l in enumerate(lines):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            body = lines[labels[m.group(1)]:i]
            n = sum('v_mfma' in x for x in body)
            if n and (best is None or n > best[0] or (n == best[0] and len(body) < len(best[1]))):
                best = (n, body)
    nv, na = g(r'; NumVgprs: (\d+)'), g(r'; NumAgprs: (\d+)')
    sc, oc = g(r'; ScratchSize: (\d+)'), g(r'; Occupancy: (\d+)')
    print(f"{dem}\n   NumVgprs {nv} NumAgprs {na} scratch {sc} occupancy {oc}")
    if best:
        n, body = best
        ins = [x.strip().split()[0] for x in body if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
        c = lambda pat: sum(bool(re.match(pat, x)) for x in ins)
        other = len(ins) - n
        print(f"   hot loop: {len(ins)} instr, mfma {n}, other/mfma {other / n:.2f} | accvgpr {c('v_accvgpr')} "
              f"valu {c('v_') - n - c('v_accvgpr')} (pk {c('v_pk_')}) ds {c('ds_')} vmem {c('global_|buffer_')} "
              f"scratch {c('scratch_')} salu {c('s_') - c('s_waitcnt') - c('s_nop') - c('s_barrier')} "
              f"waitcnt {c('s_waitcnt')} nop {c('s_nop')} barrier {c('s_barrier')}")
