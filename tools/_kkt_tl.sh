#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kkt_tl -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/kkt_tl.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls gpurun_out/kkt_tl/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY' > gpurun_out/kkt_timeline.txt
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n):
    n = re.sub(r'dotsocp::', '', n); n = re.sub(r'\(.*', '', n); n = n.replace('void ', '')
    return n[:50]
idx = [i for i, r in enumerate(rows) if 'k_kkt_cells' in r['Kernel_Name']]
for k in idx[2:5]:
    t0 = int(rows[k - 12]['Start_Timestamp']); pe = t0
    for r in rows[k - 12:k + 16]:
        s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
        print(f"{s:9.1f} {e:9.1f} {e-s:8.1f} gap {s-pe:7.1f} {short(r['Kernel_Name'])}")
        pe = e
    print('----')
PY
rm -rf gpurun_out/kkt_tl
tail -5 gpurun_out/kkt_timeline.txt
