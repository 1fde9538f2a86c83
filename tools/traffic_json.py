#!/usr/bin/env python3
"""profiles/cone_proj_traffic.json from the two per-kernel PMC tables of a round (tools/pmc_summary.py output):
`python tools/traffic_json.py fetch.csv write.csv ROUND COMMIT > profiles/cone_proj_traffic.json`.  rocprofv3 reports FETCH_SIZE /
WRITE_SIZE in KiB per dispatch; FETCH_SIZE is doubled per the gfx950 note of MI355X_MICROARCH.md."""
import csv
import json
import sys


def pick(path, col):
    for r in csv.DictReader(open(path)):
        if r["kernel"].startswith("void dotsocp::k_cone_fused<1, 4"):
            return float(r[col]) * 1024.0
    raise SystemExit("k_cone_fused<1, 4> not in " + path)


fetch = 2.0 * pick(sys.argv[1], "avg_FETCH_SIZE")
write = pick(sys.argv[2], "avg_WRITE_SIZE")
print(json.dumps({"grid": [1024, 1024, 128], "kernel": "k_cone_fused<1, 4>", "hbm_bytes_per_launch": fetch + write,
                  "fetch_bytes": fetch, "write_bytes": write,
                  "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/r%02d_pmc_*_per_kernel.csv, "
                            "KiB per dispatch); FETCH_SIZE doubled per the gfx950 note of MI355X_MICROARCH.md" % int(sys.argv[3]),
                  "round": int(sys.argv[3]), "commit": sys.argv[4]}, indent=1))
