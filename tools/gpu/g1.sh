R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4a; mkdir -p $O
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 python3 $R/tools/shape_table.py sd1 64 2 unet 3 > $O/sd15_shape_table.txt 2> $O/err1.log
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/tools/unet_eval.py sd1 64 2 5 $O/sd15_oplist.txt > $O/kt.log 2>&1 < /dev/null
find $O/kt -name "*kernel_trace.csv" -exec cp {} $O/sd15_kernel_trace.csv \;
find $O/kt -name "*kernel_stats.csv" -exec cp {} $O/sd15_kernel_stats.csv \;
rm -rf $O/kt
cd $R
timeout 400 python3 bench.py --workload sd15 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_sd15.json 2> $O/err2.log
tail -c 1500 $O/bench_sd15.json
