"""Interleaved A/B of one environment switch with bench.py: `python tools/ab.py VAR v0 v1 [v2 ...] -- [bench.py args]`.

Single bench runs on the pool's boxes drift by several percent with what ran before them; short runs
of every variant in turn, repeated, and medians per variant give differences that repeat to about 0.1 %.  Prints the
median it/s and the median of every per-phase kernel time (HIP events of the bench line) per variant."""
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    argv = sys.argv[1:]
    if "--" in argv:
        i = argv.index("--")
        argv, extra = argv[:i], argv[i + 1:]
    else:
        extra = []
    var, values = argv[0], argv[1:]
    reps = int(os.environ.get("AB_REPEATS", "5"))
    res = {v: [] for v in values}
    for _ in range(reps):
        for v in values:
            env = dict(os.environ)
            env[var] = v
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "60", "--warmup", "10"] + extra,
                                 env=env, capture_output=True, text=True, cwd=ROOT).stdout.strip().splitlines()
            if not out:
                continue
            d = json.loads(out[-1])
            res[v].append(d)
    for v in values:
        if not res[v]:
            print(var, v, "no result")
            continue
        keys = [k for k, x in res[v][0]["kernel_ms"].items() if x > 0]
        med = {k: round(statistics.median(d["kernel_ms"][k] for d in res[v]), 4) for k in keys}
        print(f"{var}={v}: it/s {statistics.median(d['value'] for d in res[v]):.2f}  {med}")


if __name__ == "__main__":
    main()
