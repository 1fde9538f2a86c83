cd $GRAFT_REPO_ROOT
B="timeout -k 10 300 python bench.py --no-cpu-baseline"
for g in "1025 1025 129" "513 513 129" "257 257 65" "129 129 33"; do $B --grid $g 2>/dev/null | python tools/benchline.py pitched; done
DOTSOCP_PITCH=0 $B --grid 1025 1025 129 2>/dev/null | python tools/benchline.py unpitched
$B 2>/dev/null | python tools/benchline.py headline
