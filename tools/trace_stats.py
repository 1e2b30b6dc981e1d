#!/usr/bin/env python
"""Per-kernel launch statistics of a `rocprofv3 --kernel-trace -f csv` run WITHOUT the warm-up
launches, for the kernels the rooflines are priced on.

    python tools/trace_stats.py <trace dir> <out.csv> <profiles/rNN_kernel_times.json> \
        --tag=c3 --steps=6 --warmup=2 "--command=python bench.py ..."

Of every kernel (by name) the FULL-SIZE launches are kept (duration above half of the longest:
a parity check on a slice of the frames is not a launch of the workload), then the first
warmup / (steps + warmup) of them -- the warm-up steps of the command -- are dropped.  Writes
<out.csv> (Name, Calls, TimedCalls, AverageNs of all / of the timed launches, Min, Max) for every
kernel above 1 % of the trace, and merges the entries of the roofline kernels into the JSON
(`<tag>_<kernel>`: avg_ms of the timed launches, launches, command), which bench.py reads for
`roofline.frac_profiled`."""
import collections, csv, glob, json, os, re, sys

KERNELS = {'llhx_kernel': 'llhx_kernel', 'lnfi_kernel': 'lnfi_kernel', 'accx_kernel': 'accx_kernel',
           'accfi_kernel': 'accf_kernel', 'accf_kernel': 'accf_kernel', 'fb_wave_kernel': 'fb_wave_kernel',
           'llh_kernel<': 'llh_kernel', 'acc_kernel<': 'acc_kernel', 'sgrad_kernel': 'sgrad_kernel',
           'accd_kernel': 'accd_kernel'}


def main():
    args = [a for a in sys.argv[1:] if not a.startswith('--')]
    opts = dict(a[2:].split('=', 1) for a in sys.argv[1:] if a.startswith('--'))
    src, out_csv, out_json = args
    tag = opts.get('tag', '')
    tag = tag + '_' if tag else ''
    steps, warmup = int(opts.get('steps', 1)), int(opts.get('warmup', 0))
    files = glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)
    f = max(files, key=os.path.getmtime)
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        per[r['Kernel_Name']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
    total = sum(d for v in per.values() for _, d in v)
    rows, entries = [], {}
    for name, v in per.items():
        v.sort()
        durs = [d for _, d in v]
        big = max(durs)
        full = [d for d in durs if d > .5 * big]
        drop = int(round(len(full) * warmup / float(steps + warmup))) if len(full) >= steps + warmup else 0
        timed = full[drop:] or full
        if sum(durs) < .01 * total:
            continue
        short = re.sub(r'\(anonymous namespace\)::|beer_mfma::|^void ', '', name)
        rows.append({'Name': short[:160], 'Calls': len(durs), 'TotalDurationNs': sum(durs),
                     'AverageNs': sum(durs) / len(durs), 'FullSizeCalls': len(full),
                     'TimedCalls': len(timed), 'TimedAverageNs': sum(timed) / len(timed),
                     'TimedMinNs': min(timed), 'TimedMaxNs': max(timed),
                     'Percentage': 100. * sum(durs) / total})
        key = next((val for k, val in KERNELS.items() if k in name), None)
        if key is not None:
            e = entries.setdefault(tag + key, {'avg_ms': 0., 'launches': 0, 'total_ms': 0.})
            # (several instantiations of one kernel family: the one with the most time stands)
            if sum(timed) / 1e6 > e['total_ms']:
                e.update(avg_ms=sum(timed) / len(timed) / 1e6, launches=len(timed),
                         total_ms=sum(timed) / 1e6, name=short[:120],
                         all_launches_avg_ms=sum(durs) / len(durs) / 1e6,
                         command=opts.get('command', ''), steps=steps, warmup=warmup)
    rows.sort(key=lambda r: -r['TotalDurationNs'])
    with open(out_csv, 'w', newline='') as g:
        w = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    old = json.load(open(out_json)) if os.path.exists(out_json) else \
        {'source': 'rocprofv3 --kernel-trace -f csv -- <command of the entry>; full-size launches, '
                   'warm-up launches dropped; see tools/trace_stats.py', 'kernels': {}}
    old['kernels'].update(entries)
    json.dump(old, open(out_json, 'w'), indent=1, sort_keys=True)
    for k, e in entries.items():
        print(f"{k:24s} {e['avg_ms']:8.3f} ms x {e['launches']}  (all launches {e['all_launches_avg_ms']:.3f})")


if __name__ == '__main__':
    main()
