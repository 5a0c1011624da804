R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4f; mkdir -p $O
cd $R; export PYTHONPATH=$R
timeout 900 python3 -m pytest tests/test_unet_gpu.py tests/test_kernels_gpu.py tests/test_determinism_gpu.py tests/test_api_gpu.py -q -m gpu -s -k "handoff or masked or survives or streaming or unet_split or layernorms_ended or fresh_processes or autotune_env or batch_slot or gguf_container" 2>&1 | grep -v "^$" | tail -40
