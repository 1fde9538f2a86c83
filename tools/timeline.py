#!/usr/bin/env python3
"""Print the kernel timeline of a few iterations from a rocprofv3 --kernel-trace csv (diagnostics)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
anchor = sys.argv[2] if len(sys.argv) > 2 else 'k_dct_axis0_pipe<false'
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 30
nit = int(sys.argv[4]) if len(sys.argv) > 4 else 2
def short(n):
    n = re.sub(r'dotsocp::', '', n); n = re.sub(r'\(.*', '', n); n = n.replace('void ', '')
    return n[:44]
starts = [i for i, r in enumerate(rows) if anchor in r['Kernel_Name']]
i0, i1 = starts[skip], starts[skip + nit]
t0 = int(rows[i0]['Start_Timestamp'])
prev_end = {}
for r in rows[max(i0 - 14, 0):i1 + 1]:
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    print(f"{s:9.1f} {e:9.1f} {e-s:8.1f} s{r['Stream_Id']} {short(r['Kernel_Name'])} grid={r['Grid_Size_X']}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}")
