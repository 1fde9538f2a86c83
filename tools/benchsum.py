#!/usr/bin/env python3
"""Print value / ms per step / box kind / phase timers of bench.py JSON lines: `python tools/benchsum.py files...`"""
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f"{f}: {d['value']:.1f} it/s {d['ms_per_step']:.3f} ms copy={d['config'].get('box_copy_gbs')} "
              f"{ {k: v for k, v in d['kernel_ms'].items() if v} }")
    except Exception as e:
        print(f, 'ERR', e, open(f).read()[-300:])
