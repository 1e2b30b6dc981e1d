#!/usr/bin/env python
"""Where a step's wall time goes: per-kernel time and idle gaps from a rocprofv3 --kernel-trace CSV.

    python tools/trace_gaps.py <dir or kernel_trace.csv> [anchor kernel substring] [n last anchors]

The trace is cut at the launches of the anchor kernel (default: the first big E-step kernel of an
iteration is found as the launch following the largest gaps); prints, for the span between the
first and the last of the last n anchors: busy time per kernel name, total idle time, and the ten
largest idle gaps with the kernels around them."""
import collections, csv, glob, os, re, sys
src = sys.argv[1]
anchor = sys.argv[2] if len(sys.argv) > 2 else 'llhx_kernel'
nlast = int(sys.argv[3]) if len(sys.argv) > 3 else 7
files = [src] if src.endswith('.csv') else glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
short = lambda n: re.sub(r'\(.*', '', n.replace('beer_mfma::(anonymous namespace)::', '').replace('(anonymous namespace)::', '').replace('void ', ''))[:70]
idx = [i for i, r in enumerate(rows) if anchor in r[2]]
idx = idx[-nlast:]
lo, hi = idx[0], idx[-1]
span = rows[lo:hi]
t0, t1 = span[0][0], rows[hi][0]
busy = collections.Counter(); calls = collections.Counter()
gaps = []
end = span[0][0]
for i, (s, e, n) in enumerate(span):
    if s > end:
        gaps.append((s - end, short(span[i - 1][2]) if i else '', short(n)))
    busy[short(n)] += e - s; calls[short(n)] += 1
    end = max(end, e)
tot = t1 - t0
print(f'span {tot / 1e6:.3f} ms over {len(idx) - 1} anchor intervals ({(tot / (len(idx) - 1)) / 1e6:.3f} ms each); '
      f'kernels {sum(busy.values()) / 1e6:.3f} ms, idle {sum(g[0] for g in gaps) / 1e6:.3f} ms in {len(gaps)} gaps')
for n, t in busy.most_common(14):
    print(f'   {t / 1e6:9.3f} ms {calls[n]:5d} x  {n}')
print('largest gaps:')
for g in sorted(gaps, reverse=True)[:10]:
    print(f'   {g[0] / 1e3:9.1f} us  after {g[1]}  before {g[2]}')
