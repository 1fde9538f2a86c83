# round-4 evidence run (one MI355X): bench lines, rocprofv3 kernel tables (1024 and 1025 grids, a rank's share), PMC traffic,
# rank shares of the 2- / 4- / 8-way split as same-run pairs with the full grid, loop variants, end-to-end multilevel solves
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ev4
mkdir -p $O
B="timeout -k 10 400 python bench.py"
$B > $O/r04_bench_default.json 2> $O/err.txt; echo default done
$B --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_bench_driver_window.json 2>> $O/err.txt
for n in 8 4 2; do $B --no-cpu-baseline --rank-share $n > $O/r04_rank_share_n$n.json 2>> $O/err.txt; done
for g in "256 256 64" "512 512 128" "1025 1025 129" "513 513 129" "257 257 65" "129 129 33" "2048 2048 32"; do
  n=$(echo $g | tr ' ' 'x'); $B --no-cpu-baseline --grid $g > $O/r04_bench_$n.json 2>> $O/err.txt; done
$B --no-cpu-baseline --workload wdot2d > $O/r04_bench_wdot2d.json 2>> $O/err.txt
$B --no-cpu-baseline --workload dot1d > $O/r04_bench_dot1d.json 2>> $O/err.txt
$B --no-cpu-baseline --workload dot1d --grid 1025 1 33 > $O/r04_bench_dot1d_1025x33.json 2>> $O/err.txt
$B --no-cpu-baseline --method PALM > $O/r04_bench_palm.json 2>> $O/err.txt
$B --no-cpu-baseline --method PALM --nslabs 2 > $O/r04_bench_palm_nslabs2.json 2>> $O/err.txt
$B --no-cpu-baseline --method acc-ADMM > $O/r04_bench_accadmm.json 2>> $O/err.txt
$B --no-cpu-baseline --nslabs 2 > $O/r04_bench_nslabs2.json 2>> $O/err.txt
$B --no-cpu-baseline --nslabs 8 > $O/r04_bench_nslabs8.json 2>> $O/err.txt
DOTSOCP_TSOLVE=dct $B --no-cpu-baseline --grid 1025 1025 129 > $O/r04_bench_1025x1025x129_transform_t_pass.json 2>> $O/err.txt
DOTSOCP_OVERLAP=0 $B --no-cpu-baseline --rank-share 8 > $O/r04_rank_share_n8_messages_on_main_stream.json 2>> $O/err.txt
echo benches done
cd /tmp && export TMPDIR=/tmp
R="timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv"
$R -d $O/prof_a -o r04a -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/r04_a_bench_under_rocprof.json 2>> $O/err.txt
$R -d $O/prof_b -o r04b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r04_b_bench_driver_window_under_rocprof.json 2>> $O/err.txt
$R -d $O/prof_c -o r04c -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --grid 1025 1025 129 > $O/r04_c_bench_1025_under_rocprof.json 2>> $O/err.txt
$R -d $O/prof_d -o r04d -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --rank-share 8 > $O/r04_d_rank_share_n8_under_rocprof.json 2>> $O/err.txt
echo traces done
P="timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv"
$P --pmc FETCH_SIZE -d $O/pmc_fetch -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>> $O/err.txt
$P --pmc WRITE_SIZE -d $O/pmc_write -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > /dev/null 2>> $O/err.txt
$P --pmc FETCH_SIZE -d $O/pmc_fetch1025 -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --grid 1025 1025 129 > /dev/null 2>> $O/err.txt
$P --pmc WRITE_SIZE -d $O/pmc_write1025 -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline --grid 1025 1025 129 > /dev/null 2>> $O/err.txt
echo pmc done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/pmc_fetch FETCH_SIZE > $O/r04_pmc_fetch_size_per_kernel.csv
python tools/pmc_summary.py $O/pmc_write WRITE_SIZE > $O/r04_pmc_write_size_per_kernel.csv
python tools/pmc_summary.py $O/pmc_fetch1025 FETCH_SIZE > $O/r04_pmc_fetch_size_per_kernel_1025x1025x129.csv
python tools/pmc_summary.py $O/pmc_write1025 WRITE_SIZE > $O/r04_pmc_write_size_per_kernel_1025x1025x129.csv
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_fetch1025 $O/pmc_write1025
for d in a b c d; do cp $O/prof_$d/*kernel_stats.csv $O/r04_${d}_kernel_stats.csv; done
rm -rf $O/prof_a $O/prof_b $O/prof_c $O/prof_d
timeout -k 10 600 python demos/multilevel_large.py 257 65 3 513 129 4 1025 129 4 > $O/r04_multilevel_large.log 2>> $O/err.txt
timeout -k 10 900 python tools/parity_fullsize.py 4 1024 128 inPALM > $O/r04_parity_1024x1024x128_inPALM_K4.log 2>&1
timeout -k 10 900 python tools/parity_fullsize.py 4 1025 129 inPALM > $O/r04_parity_1025x1025x129_inPALM_K4.log 2>&1
tail -6 $O/r04_parity_1024x1024x128_inPALM_K4.log
ls $O | head -80; tail -5 $O/err.txt
