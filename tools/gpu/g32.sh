R=$GRAFT_REPO_ROOT; cd $R; export PYTHONPATH=$R
bash tools/profile_round.sh sd15 1 r4 2>&1 | tail -3
bash tools/profile_round.sh sdxl 4 r4 2>&1 | tail -3
O=$R/gpurun_out/profiles_r4
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt2 -- python3 $R/tools/unet_eval.py sd1 64 2 5 $O/sd15_oplist.txt > $O/kt2.log 2>&1 < /dev/null
find $O/kt2 -name "*kernel_trace.csv" -exec cp {} $O/sd15_kernel_trace.csv \;
rm -rf $O/kt2
python3 $R/tools/trace_join.py $O/sd15_kernel_trace.csv $O/sd15_oplist.txt > $O/r4_sd15_b1_trace_shape_table_after.txt 2>&1
rm -f $O/sd15_kernel_trace.csv
cd $R
timeout 1500 python3 bench.py --steps 5 --warmup 2 > $O/r4_bench_default_line.json 2> $O/bench_err.log; tail -c 300 $O/r4_bench_default_line.json
ls $O
