R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4h; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 900 python3 -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "split_k" 2>&1 | tail -8
for e in 1 0; do MLSD_NO_SPLITK_PAR=$e timeout 200 python3 tools/two_stream_eval.py sd1 64 2 20 8 2>&1 | tail -1; done
MLSD_NO_SPLITK_PAR=1 timeout 200 python3 tools/two_stream_eval.py sd1 64 2 20 8 2>&1 | tail -1
timeout 200 python3 tools/two_stream_eval.py sd1 64 2 20 8 2>&1 | tail -1
timeout 600 python3 -m pytest tests/test_golden_gpu.py tests/test_determinism_gpu.py -x -q -m gpu -k "sd15 or sd1 or tiny" 2>&1 | tail -4
