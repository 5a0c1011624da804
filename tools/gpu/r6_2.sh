# round 6, call 2: the small-Cout streaming convolution -- parity, microbench, decoders in the plan
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
( timeout 900 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv2d_small_n or test_conv2d" 2>&1 | tail -15 ) > gpurun_out/r6/t2_conv_smalln_tests.txt
tail -5 gpurun_out/r6/t2_conv_smalln_tests.txt
timeout 600 python tools/conv_smalln_bench.py 10 > gpurun_out/r6/conv_smalln_bench.txt 2>&1; tail -40 gpurun_out/r6/conv_smalln_bench.txt
( timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_pipeline_gpu.py -x -q -k "vae or tae or decode" 2>&1 | tail -8 ) > gpurun_out/r6/t2_decode_tests.txt
tail -4 gpurun_out/r6/t2_decode_tests.txt
timeout 300 python3 tools/shape_table.py sdxl 128 4 vae > gpurun_out/r6/vae_shape_smalln.txt 2>&1; head -3 gpurun_out/r6/vae_shape_smalln.txt; grep "conv3x3n16\|x3x" gpurun_out/r6/vae_shape_smalln.txt
timeout 300 python3 tools/shape_table.py sdxl 128 4 tae > gpurun_out/r6/tae_shape_smalln.txt 2>&1; head -3 gpurun_out/r6/tae_shape_smalln.txt; grep "conv3x3n16\|x3x" gpurun_out/r6/tae_shape_smalln.txt
MLSD_CONV_SMALLN=0 timeout 300 python3 tools/shape_table.py sdxl 128 4 tae > gpurun_out/r6/tae_shape_gemm.txt 2>&1; head -3 gpurun_out/r6/tae_shape_gemm.txt; grep "x3x" gpurun_out/r6/tae_shape_gemm.txt
