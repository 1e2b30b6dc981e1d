#!/usr/bin/env python
"""Per-kernel means of a rocprofv3 --pmc pass (counter_collection.csv), full-size launches only.

    python tools/pmc_quick.py <dir or csv> [kernel substring ...]
"""
import collections, csv, glob, os, re, sys
src = sys.argv[1]
pats = sys.argv[2:] or ['llhx_kernel', 'accx_kernel', 'accfi_kernel', 'accf_kernel', 'fb_wave_kernel', 'llh_kernel<', 'acc_kernel<']
files = [src] if src.endswith('.csv') else glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True)
per = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        name = r['Kernel_Name']
        key = next((p for p in pats if p in name), None)
        if key is None:
            continue
        short = name.replace('beer_mfma::(anonymous namespace)::', '').replace('void ', '')
        short = re.sub(r'\(.*', '', short)
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
        per[short][r['Counter_Name']].append((float(r['Counter_Value']), dur))
for k, counters in per.items():
    print(k)
    for name, vals in sorted(counters.items()):
        big = max(t for _, t in vals)
        full = [(v, t) for v, t in vals if t > .5 * big]
        print(f'   {name:32s} {sum(v for v, _ in full) / len(full):16.4g}   ({len(full)} launches, {sum(t for _, t in full) / len(full):.3f} ms)')
