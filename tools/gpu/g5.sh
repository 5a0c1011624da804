R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4e; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 600 python3 -m pytest tests/test_unet_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -s -k "handoff or masked or f16_table or survives" 2>&1 | tail -15
timeout 900 python3 tools/parity_f16_tables.py > $O/r4_parity_f16_tables.txt 2> $O/parity_err.log; cat $O/r4_parity_f16_tables.txt; tail -3 $O/parity_err.log
timeout 1500 python3 -m pytest tests -q -m gpu --durations=40 -p no:cacheprovider > $O/pytest_full.log 2>&1; tail -60 $O/pytest_full.log
