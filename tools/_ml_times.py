import os, sys, time
sys.path.insert(0, os.getcwd())
import dotsocp_amd as D
from dotsocp_amd import solvers as S
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
D.solver_dotsocp2d(*D.get_example_2d("example1", 33, 33), 9, 1, dict(tol=1e-2, maxit=20), "inPALM")
orig_run, orig_close = S.InPALMContext.run, S.InPALMContext.close
kept = []
def run(self, n=-1):
    if mode == "sleep":
        time.sleep(0.3)
    self.synchronize()
    parts = []
    for k in [1] * 24 + [10**9]:
        a = time.perf_counter(); d = orig_run(self, k); self.synchronize(); parts.append(round((time.perf_counter() - a) * 1e3, 1))
        if d < k: break
    print("  ms per piece", parts)
    return 0
def close(self):
    if mode == "keep" and getattr(self, "_ctx", None):
        kept.append(self._ctx); self._ctx = None
    else:
        orig_close(self)
S.InPALMContext.run = run
S.InPALMContext.close = close
rho0, rho1 = D.get_example_2d("example1", 513, 513)
for rep in range(2):
    t = time.perf_counter()
    out, timeML, histML, hist = D.solver_dotsocp2d(rho0, rho1, 129, 4, dict(tol=1e-4, maxit=3000), "inPALM")
    print(mode, "wall", round(time.perf_counter() - t, 3), [round(float(x["Total_Time"]), 3) for x in timeML[:-1]])
