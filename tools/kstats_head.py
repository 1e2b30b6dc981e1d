#!/usr/bin/env python
"""Compact top-N of a rocprofv3 kernel_stats.csv:  python tools/kstats_head.py <csv or dir> [N]"""
import csv, glob, os, re, sys
src = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = src if src.endswith('.csv') else max(glob.glob(os.path.join(src, '**', '*kernel_stats.csv'), recursive=True),
                                          key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f'{f}: {tot / 1e6:.2f} ms in {len(rows)} kernels')
for r in rows[:n]:
    name = re.sub(r'^void ', '', r['Name'])
    name = re.sub(r'\(anonymous namespace\)::|beer_mfma::|at::native::', '', name)
    name = re.sub(r'\(.*', '', name)[:80]
    print(f"{int(r['TotalDurationNs']) / 1e6:9.2f} ms {r['Percentage']:>6}% {r['Calls']:>6} x {float(r['AverageNs']) / 1e6:8.3f} ms  {name}")
