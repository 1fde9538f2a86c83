#!/usr/bin/env python3
"""One-line digest of bench.py's JSON line (stdin): grid, it/s, roofline fraction, phase timers."""
import json
import sys

d = json.loads(sys.stdin.read())
tag = sys.argv[1] if len(sys.argv) > 1 else ""
print(tag, d["config"]["grid"], round(d["value"], 1), "it/s", round(d["ms_per_step"], 3), "ms  frac", round(d["roofline"]["frac"], 3),
      {k: v for k, v in d["kernel_ms"].items() if v})
