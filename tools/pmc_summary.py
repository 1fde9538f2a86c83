#!/usr/bin/env python3
"""Per-kernel average of one rocprofv3 PMC counter.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 bench.py ...
    python tools/pmc_summary.py out FETCH_SIZE > profiles/rNN_pmc_fetch_size_per_kernel.csv

Reads every *counter_collection.csv below the directory and prints `kernel,launches,avg_value` (the counter
summed over the XCD / channel instances of a dispatch, then averaged over the dispatches of a kernel).
FETCH_SIZE / WRITE_SIZE are in KiB... no: in units of 1 KB as rocprofv3 reports them; the gfx950 correction of
MI355X_MICROARCH.md (FETCH_SIZE counts 1/2 of 8- and 16-byte-per-lane streaming loads) is applied by the reader,
not here."""
import collections
import csv
import glob
import os
import sys


def main():
    root, counter = sys.argv[1], sys.argv[2]
    per_dispatch = collections.defaultdict(float)
    name_of = {}
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            key = (f, r["Dispatch_Id"])
            per_dispatch[key] += float(r["Counter_Value"])
            name_of[key] = r["Kernel_Name"]
    acc = collections.defaultdict(list)
    for key, v in per_dispatch.items():
        acc[name_of[key]].append(v)
    w = csv.writer(sys.stdout)
    w.writerow(["kernel", "launches", f"avg_{counter}"])
    for k, vs in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, len(vs), sum(vs) / len(vs)])


if __name__ == "__main__":
    main()
