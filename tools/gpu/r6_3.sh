# round 6, call 3: measured parity figures for the tolerance split, the attention fill sweep, conv_smalln defaults, the whole GPU suite
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6
timeout 300 python tools/attn_fill.py 30 > gpurun_out/r6/attn_fill.txt 2>&1; cat gpurun_out/r6/attn_fill.txt | tail -12
timeout 600 python tools/conv_smalln_bench.py 10 > gpurun_out/r6/conv_smalln_bench.txt 2>&1; grep "automatic\|implicit" gpurun_out/r6/conv_smalln_bench.txt
( timeout 1500 python -m pytest tests/test_golden_gpu.py tests/test_unet_gpu.py -q -s 2>&1 | grep -E "rel-L2|unet_|vae_|tae_|clip_|gen_|passed|failed|HIP vs" ) > gpurun_out/r6/parity_figures.txt
tail -60 gpurun_out/r6/parity_figures.txt
( timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 ) > gpurun_out/r6/gpu_suite_a.txt
tail -5 gpurun_out/r6/gpu_suite_a.txt
