import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
import dotsocp_amd as D
rho0, rho1 = D.get_example_2d("example1", 513, 513)
out, timeML, histML, hist = D.solver_dotsocp2d(rho0, rho1, 129, 4, dict(tol=1e-4, maxit=3000), "inPALM")
out, timeML, histML, hist = D.solver_dotsocp2d(rho0, rho1, 129, 4, dict(tol=1e-4, maxit=3000), "inPALM")
print([round(float(x["Total_Time"]), 3) for x in timeML[:-1]])
