#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-trace --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/mltr -- python3 $GRAFT_REPO_ROOT/tools/_ml_trace.py > $GRAFT_REPO_ROOT/gpurun_out/mltr.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/mltr.log
k=$(ls gpurun_out/mltr/*/*kernel_trace.csv | head -1); h=$(ls gpurun_out/mltr/*/*hip_api_trace.csv | head -1)
m=$(ls gpurun_out/mltr/*/*memory_copy_trace.csv | head -1)
python3 - "$k" "$h" "$m" <<'PY'
import csv, sys, re
K = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
H = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r['Start_Timestamp']))
M = sorted(csv.DictReader(open(sys.argv[3])), key=lambda r: int(r['Start_Timestamp']))
print("memcopy columns", list(M[0].keys()) if M else None, len(M))
# largest gaps between consecutive kernels in the last 40 % of the run
n = len(K); gaps = []
for i in range(int(n * 0.55), n - 1):
    g = int(K[i + 1]['Start_Timestamp']) - int(K[i]['End_Timestamp'])
    gaps.append((g, i))
gaps.sort(reverse=True)
for g, i in gaps[:2]:
    a, b = int(K[i]['End_Timestamp']), int(K[i + 1]['Start_Timestamp'])
    print(f"gap {g/1e6:.1f} ms after {K[i]['Kernel_Name'][:60]} before {K[i+1]['Kernel_Name'][:60]}")
    for r in M:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if e > a - 20000000 and s < b:
            print(f"    COPY {r.get('Direction')} {r.get('Bytes', r.get('Size'))} B: {(e - s)/1e6:.3f} ms (starts {(s - a)/1e6:.3f} ms into the gap)")
    for r in H:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if e > a and s < b and e - s > 100000:
            print(f"    {r['Function']}: {(e - s)/1e6:.3f} ms (starts {(s - a)/1e6:.3f} ms into the gap) tid {r.get('Thread_Id')}")
PY
rm -rf gpurun_out/mltr
