"""oracle/ -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

CPU restatement (numpy + a small plain-C file) of the inPALM/ADMM SOCP iteration
loop of chlhnu/DOT-SOCP (socp/dot1d, socp/dot2d, socp/wdot2d), each function citing
the reference file:line it follows.

PARITY UNPINNED: the reference contains no tests, fixtures or golden vectors for
this path, MATLAB/Octave are not available, and the prebuilt MEX binaries that ship
inside the reference are never loaded or executed.  The restatement is therefore
pinned only by (i) algebraic invariants that any faithful implementation must obey
and (ii) cross-checks against the known answers recorded in SURVEY.md section 8c(iii).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package -- as the checker / the reported CPU baseline, never as the product path.
"""
