import os, sys, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import dotsocp_amd as D
n, nt, L = 1025, 129, 4
rho0, rho1 = D.get_example_2d("example1", n, n)
D.solver_dotsocp2d(*D.get_example_2d("example1", 65, 65), 17, 2, dict(tol=1e-3, maxit=100), "inPALM")  # warm the runtime
pr = cProfile.Profile()
t = time.perf_counter()
pr.enable()
out, timeML, histML, hist = D.solver_dotsocp2d(rho0, rho1, nt, L, dict(tol=1e-4, maxit=3000), "inPALM")
pr.disable()
print("wall", time.perf_counter() - t, [round(float(x["Total_Time"]), 2) for x in timeML[:-1]])
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
