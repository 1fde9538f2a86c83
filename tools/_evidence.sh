# round-2 evidence run (one MI355X): bench lines, rocprofv3 kernel table, PMC traffic, full-size parity logs
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/ev
mkdir -p $O
B="timeout -k 10 400 python bench.py"
$B > $O/r02_bench_default.json 2> $O/err.txt; echo default done
$B --steps 20 --warmup 5 --no-cpu-baseline > $O/r02_bench_driver_window.json 2>> $O/err.txt
$B --no-cpu-baseline --grid 256 256 64 > $O/r02_bench_256x256x64.json 2>> $O/err.txt
$B --no-cpu-baseline --grid 512 512 128 > $O/r02_bench_512x512x128.json 2>> $O/err.txt
$B --no-cpu-baseline --workload wdot2d > $O/r02_bench_wdot2d.json 2>> $O/err.txt
$B --no-cpu-baseline --workload dot1d > $O/r02_bench_dot1d.json 2>> $O/err.txt
$B --no-cpu-baseline --grid 1025 1025 129 --steps 60 > $O/r02_bench_1025x1025x129.json 2>> $O/err.txt
$B --no-cpu-baseline --grid 257 257 65 > $O/r02_bench_257x257x65.json 2>> $O/err.txt
$B --no-cpu-baseline --method PALM > $O/r02_bench_palm.json 2>> $O/err.txt
$B --no-cpu-baseline --method acc-ADMM > $O/r02_bench_accadmm.json 2>> $O/err.txt
$B --no-cpu-baseline --nslabs 2 > $O/r02_bench_nslabs2.json 2>> $O/err.txt
$B --no-cpu-baseline --nslabs 8 > $O/r02_bench_nslabs8.json 2>> $O/err.txt
$B --no-cpu-baseline --grid 2048 2048 32 > $O/r02_bench_2048x2048x32.json 2>> $O/err.txt
$B --no-cpu-baseline --grid 2048 2048 32 --nslabs 2 > $O/r02_bench_2048x2048x32_nslabs2.json 2>> $O/err.txt
echo benches done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_a -o r02a -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/r02_a_bench_under_rocprof.json 2>> $O/err.txt
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b -o r02b -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/r02_b_bench_driver_window_under_rocprof.json 2>> $O/err.txt
echo traces done
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/pmc_fetch_bench.json 2>> $O/err.txt
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $O/pmc_write_bench.json 2>> $O/err.txt
echo pmc done
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $O/pmc_fetch FETCH_SIZE > $O/r02_pmc_fetch_size_per_kernel.csv
python tools/pmc_summary.py $O/pmc_write WRITE_SIZE > $O/r02_pmc_write_size_per_kernel.csv
find $O/prof_a $O/prof_b -name "*kernel_stats.csv" | head
timeout -k 10 900 python tools/parity_fullsize.py 4 1024 128 inPALM > $O/r02_parity_1024x1024x128_inPALM_K4.log 2>&1
timeout -k 10 400 python tools/parity_fullsize.py 3 512 128 PALM > $O/r02_parity_512x512x128_PALM_K3.log 2>&1
timeout -k 10 400 python tools/parity_fullsize.py 3 512 128 acc-ADMM > $O/r02_parity_512x512x128_accADMM_K3.log 2>&1
tail -8 $O/r02_parity_1024x1024x128_inPALM_K4.log
rm -rf $O/pmc_fetch $O/pmc_write
ls -la $O
