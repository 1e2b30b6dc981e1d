#!/usr/bin/env python
"""Summarise rocprofv3 PMC passes of `bench.py` into profiles/<round>_pmc.json.

    python tools/pmc_summary.py r02 [--command='python bench.py ...'] [--tag=c3] \
        gpurun_out/pmc_a gpurun_out/pmc_b gpurun_out/pmc_c

Each directory holds one `rocprofv3 --kernel-trace --pmc ...` pass.  Only the
full-size launches (1 M frames) of the four E-step kernels are kept; the
filtered rows are written next to the summary as <round>_pmc_pass<i>.csv.
GRBM_GUI_ACTIVE is summed over the 8 XCDs; FETCH_SIZE / WRITE_SIZE are KiB as
reported (MI355X_MICROARCH.md: FETCH_SIZE under-reports 16 B/lane coalesced
reads by 2x; corrected where bench.py uses it, raw here).
"""

import collections
import csv
import glob
import json
import os
import sys

KERNELS = {'llhx_kernel': 'llhx_kernel', 'lnfi_kernel': 'lnfi_kernel', 'accx_kernel': 'accx_kernel',
           'accfi_kernel': 'accf_kernel', 'accf_kernel': 'accf_kernel', 'fb_wave_kernel': 'fb_wave_kernel',
           'llh_kernel<': 'llh_kernel', 'acc_kernel<': 'acc_kernel', 'sgrad_kernel': 'sgrad_kernel',
           'accd_kernel': 'accd_kernel'}
# kernels whose streaming reads are 16 B per lane: FETCH_SIZE counts half their bytes on
# gfx950 (MI355X_MICROARCH.md, HBM section); bench.py doubles the read figure for these
WIDE_LOADS = {'llhx_kernel': True, 'lnfi_kernel': True, 'accx_kernel': True, 'accf_kernel': True,
              'fb_wave_kernel': False, 'llh_kernel': True, 'acc_kernel': True, 'sgrad_kernel': False,
              # (4 B per lane, but 64 consecutive lanes: 128-byte requests all the same -- calibrated on
              #  its operands, round 5: 0.368 GB counted for the 0.736 GB of weights + samples it reads)
              'accd_kernel': True}
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rnd, dirs = sys.argv[1], sys.argv[2:]
    command = 'python bench.py --steps 3 --warmup 1 --no-cpu-baseline'
    if dirs and dirs[0].startswith('--command='):
        command, dirs = dirs[0][len('--command='):], dirs[1:]
    tag = ''
    if dirs and dirs[0].startswith('--tag='):
        tag, dirs = dirs[0][len('--tag='):] + '_', dirs[1:]
    # --skip=W/N: the first W of every N full-size launches of a kernel are warm-up steps
    skip = (0, 1)
    if dirs and dirs[0].startswith('--skip='):
        a, b = dirs[0][len('--skip='):].split('/')
        skip, dirs = (int(a), int(b)), dirs[1:]
    # --min-read=kernel:bytes,...: the bytes a launch of a kernel cannot avoid reading (its operands,
    # once).  A counter read below that is the halved tally of 128-byte requests (the guide's
    # calibration rule: "calibrate on a known byte count in your own access pattern"): the entry is
    # flagged `read_below_operands` and its `wide_loads` set, whatever the table above says.
    min_read = {}
    if dirs and dirs[0].startswith('--min-read='):
        for item in dirs[0][len('--min-read='):].split(','):
            k_, v_ = item.split(':')
            min_read[k_] = float(v_)
        dirs = dirs[1:]
    out = collections.defaultdict(dict)
    for i, d in enumerate(dirs, start=1):
        # (gpurun merges a new run's files into the local directory: take the newest)
        f = max(glob.glob(os.path.join(d, '*', '*counter_collection.csv')) +
                glob.glob(os.path.join(d, '*counter_collection.csv')), key=os.path.getmtime)
        rows = list(csv.DictReader(open(f)))
        keep = []
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            key = next((v for k, v in KERNELS.items() if k in r['Kernel_Name']), None)
            if key is None:
                continue
            keep.append(r)
            dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
            per[key][r['Counter_Name']].append((int(r['Start_Timestamp']), float(r['Counter_Value']), dur))
        for key, counters in per.items():
            for name, vals in counters.items():
                vals = [(v, t) for _, v, t in sorted(vals)]
                big = max(v for v, _ in vals)
                full = [(v, t) for v, t in vals if v > .5 * big] or vals     # full-size launches
                drop = int(round(len(full) * skip[0] / float(skip[1])))
                full = full[drop:] or full                                   # (warm-up steps dropped)
                out[key][name] = sum(v for v, _ in full) / len(full)
                out[key][name + '_ms'] = sum(t for _, t in full) / len(full)
        with open(os.path.join(ROOT, 'profiles', f'{rnd}_pmc_{tag}pass{i}.csv'), 'w', newline='') as g:
            w = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(keep)
    for key, k in out.items():
        if 'SQ_VALU_MFMA_BUSY_CYCLES' in k and 'GRBM_GUI_ACTIVE' in k:
            cycles = k['GRBM_GUI_ACTIVE'] / 8.                   # per XCD
            k['mfma_util'] = k['SQ_VALU_MFMA_BUSY_CYCLES'] / (cycles * 1024)
            k['clock_ghz'] = cycles / (k['GRBM_GUI_ACTIVE_ms'] * 1e6)
        if 'FETCH_SIZE' in k:
            k['hbm_read_bytes_raw'] = k['FETCH_SIZE'] * 1024
        if 'WRITE_SIZE' in k:
            k['hbm_write_bytes'] = k['WRITE_SIZE'] * 1024
        k['wide_loads'] = WIDE_LOADS.get(key, False)
        if key in min_read and 'hbm_read_bytes_raw' in k:
            k['operand_read_bytes'] = min_read[key]
            if k['hbm_read_bytes_raw'] < .75 * min_read[key]:
                k['read_below_operands'] = True
                k['wide_loads'] = True
                print(f'{tag}{key}: FETCH_SIZE {k["hbm_read_bytes_raw"] / 1e9:.3f} GB is below the '
                      f'{min_read[key] / 1e9:.3f} GB of operands: counted at half, doubled from here on')
        k['command'] = command
    path = os.path.join(ROOT, 'profiles', f'{rnd}_pmc.json')
    old = json.load(open(path))['kernels'] if os.path.exists(path) else {}
    old.update({tag + key: k for key, k in out.items()})     # 'c3_accf_kernel', ...
    json.dump({'source': 'rocprofv3 --kernel-trace --pmc (separate passes) -- <command of the entry>; '
                         'full-size launches only, averaged per launch; see tools/pmc_summary.py',
               'kernels': old},
              open(path, 'w'), indent=1)
    for key, k in out.items():
        print(key, {n: round(v, 4) for n, v in k.items() if n in ('mfma_util', 'clock_ghz')},
              k.get('hbm_read_bytes_raw'), k.get('hbm_write_bytes'))


if __name__ == '__main__':
    main()
