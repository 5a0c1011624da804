R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4u; mkdir -p $O; cd $R; export PYTHONPATH=$R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "column_statistics or groupnorm or split_k" 2>&1 | tail -5
for e in 1 0; do echo "MLSD_GN_TWO_PASS=$e"; MLSD_GN_TWO_PASS=$e timeout 300 python3 tools/quick_decode_perf.py sdxl 128 4 2>&1 | grep -E "^decode|groupnorm|ops=" | head -14; MLSD_GN_TWO_PASS=$e timeout 300 python3 tools/quick_decode_perf.py sd1 64 1 2>&1 | grep -E "^decode"; done 2>&1 | tee $O/gn_stats_ab.txt
for e in 1 0; do echo "MLSD_GN_TWO_PASS=$e"; MLSD_GN_TWO_PASS=$e timeout 300 python3 tools/unet_eval.py sd1 64 2 30 2>&1 | tail -1; MLSD_GN_TWO_PASS=$e timeout 300 python3 tools/unet_eval.py sdxl 128 8 8 2>&1 | tail -1; done 2>&1 | tee -a $O/gn_stats_ab.txt
