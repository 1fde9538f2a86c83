cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/dp
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 30 --warmup 5 > $O/bench.json 2> $O/err.txt
find $O/prof -name "*kernel_stats.csv" -exec cp {} $O/pipe_kernel_stats.csv \;
rm -rf $O/prof
grep -i "dct" $O/pipe_kernel_stats.csv | awk -F'",' '{print substr($1,1,60), $2}' | cut -c1-95
