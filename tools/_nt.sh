cd $GRAFT_REPO_ROOT
for v in 1 3 5 7 1 3 5 7 0; do DOTSOCP_NT=$v timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python tools/benchline.py nt$v; done
