#!/bin/bash
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/j34; mkdir -p $O
rocprofv3 --kernel-trace -f csv -d $O/tr -- python bench.py --no-cpu-baseline --no-exact --no-check --steps 3 --warmup 2 > $O/c2.json 2>$O/err.log
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/j34/tr/*/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
# last step: find the last llh16 launch, print from the previous acc16d end to the end
idx = [i for i, r in enumerate(rows) if 'llh16_kernel' in r['Kernel_Name']]
lo, hi = idx[-2], idx[-1]
prev_end = None
out = open('gpurun_out/j34/last_step.txt', 'w')
for r in rows[lo:hi + 1]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = 0 if prev_end is None else (s - prev_end) / 1e3
    out.write(f"{gap:8.1f} gap  {(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:110]}\n")
    prev_end = e
out.close()
PY
find $O/tr -name '*.csv' -delete
cat gpurun_out/j34/last_step.txt | cut -c1-150
