R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4b; mkdir -p $O
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 $R/tools/unet_eval.py sd1 64 2 5 $O/sd15_oplist.txt > $O/kt.log 2>&1 < /dev/null
find $O/kt -name "*kernel_trace.csv" -exec cp {} $O/sd15_kernel_trace.csv \;
rm -rf $O/kt
python3 $R/tools/trace_join.py $O/sd15_kernel_trace.csv $O/sd15_oplist.txt > $O/sd15_trace_shape_table.txt 2>&1
cd $R
timeout 900 python3 -m pytest tests/test_golden_gpu.py -x -q -m gpu -k "bench_plan or config3" -s 2>&1 | tail -25 > $O/pytest.log
tail -25 $O/pytest.log
