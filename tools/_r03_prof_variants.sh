cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/variants
mkdir -p $O
for m in PALM acc-ADMM; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$m -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --method $m --steps 60 --warmup 10 > $O/$m.json 2>/dev/null
python3 - <<PY
import csv
print("== $m")
tot=0
for r in csv.DictReader(open("$O/$m/p_kernel_stats.csv")):
    if float(r["Percentage"]) > 0.4: print("%-110s %5s %9.1f us %6.2f%%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"])/1e3, float(r["Percentage"])))
PY
python3 $GRAFT_REPO_ROOT/tools/benchline.py $m < $O/$m.json
done
